"""Multi-GPU layout of the path: one process per GPU, clips sharded statically, weights replicated.

The reference has no multi-GPU inference (Lightning DDP is only used for training), and
the path shards into independent units -- clips (and, inside a clip, windows) -- so there is NO
data-path collective.  The only exchange is the end-of-run reduction of the metric histograms
(int64[3,K] + frame count, SUM) and of the elapsed time (MAX): ~130 bytes over RCCL (backend "nccl"
on ROCm) or gloo on CPU.
"""
import os

import torch
import torch.distributed as dist


def world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process when absent)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_group(backend, rank, local_rank, size):
    """dist.init_process_group for this path.  RCCL ("nccl"): the rank's GPU is made current first AND named as `device_id`, which
    binds the process group to that device and creates its communicator eagerly -- every later collective and barrier runs on
    this GPU without any device guessing, and an IPC / RCCL set-up failure surfaces HERE, before any work (what bench.py's
    launcher looks for on stderr), not inside the first collective at the end of the run."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=size, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=size)


def init(backend=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1; returns (rank, local_rank, world_size)."""
    rank, local_rank, size = world()
    if size > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        init_group(backend, rank, local_rank, size)
    return rank, local_rank, size


def clips_for_rank(num_clips, rank, world_size):
    """Static round-robin: rank r takes clips r, r+W, ... (all clips cost the same: 21 frames each)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_clips, world_size))


def windows_of_clip(num_frames, frame_delta):
    """Key-frame windows of one clip as FlowData(split='predict') enumerates them
    (flow/dataset.py:64,113-114): window i spans frames [i*d, (i+1)*d], num_frames // d windows."""
    return [(i * frame_delta, (i + 1) * frame_delta) for i in range(num_frames // frame_delta)]


def clip_window_schedule(num_clips, num_frames, frame_delta, rank, world_size):
    """The (clip, key_prev, key_next) windows rank r processes, in order, when `num_clips` clips are sharded by clip
    (BASELINE configs[4]: 64 clips of 21 frames, frame_delta 5 -> 4 windows per clip): clips r, r+W, ... each walked window by
    window as FlowData(split='predict') enumerates them.  Over all ranks every (clip, window) appears exactly once."""
    return [(c, k0, k1) for c in clips_for_rank(num_clips, rank, world_size) for (k0, k1) in windows_of_clip(num_frames, frame_delta)]


def window_block(num_windows, rank, world_size):
    """Frame-window sharding of ONE long clip: rank r takes the contiguous block
    [num_windows*r // W, num_windows*(r+1) // W) -- blocks differ by at most one window and may be empty when W > windows."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return range(num_windows * rank // world_size, num_windows * (rank + 1) // world_size)


def barrier(device=None):
    """dist.barrier; on RCCL the rank's own device is named explicitly (device_ids) so that the barrier's internal
    all-reduce can never run on another rank's GPU."""
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            idx = torch.device(device).index if device is not None else None
            dist.barrier(device_ids=[idx if idx is not None else torch.cuda.current_device()])  # device="cuda" has no index
        else:
            dist.barrier()


def describe():
    """What the launcher really set up, for the benchmark's JSON line: (backend name, world size as torch.distributed sees it)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_backend(), dist.get_world_size()
    return "none", 1


def exchange_boundary(last_mask, has_windows, device="cpu"):
    """Frame-window sharding drops nothing: the temporal-consistency metric (flow/base.py:280-295) pairs every frame with its
    predecessor, and the predecessor of a rank's FIRST frame is the LAST frame of the nearest earlier rank that had windows.
    Every rank contributes its last uint8 mask [H,W] (zeros + has_windows=False when its block is empty); returns the mask to
    pair this rank's first frame with, or None (rank 0 / no earlier rank had windows / single process).  One all_gather of
    H*W bytes per rank at the end of the run -- not in the per-window loop."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    rank, size = dist.get_rank(), dist.get_world_size()
    mine = torch.as_tensor(last_mask, dtype=torch.uint8, device=device).contiguous()
    flag = torch.tensor([1 if has_windows else 0], dtype=torch.uint8, device=device)
    masks = [torch.empty_like(mine) for _ in range(size)]
    flags = [torch.empty_like(flag) for _ in range(size)]
    dist.all_gather(masks, mine)
    dist.all_gather(flags, flag)
    if not has_windows:
        return None
    for q in range(rank - 1, -1, -1):
        if int(flags[q].item()):
            return masks[q]
    return None


def reduce_run(hist, frames, seconds, device="cpu"):
    """Combine per-rank results: (sum of int64[3,K] histograms, total frames, max seconds).  `device`: where the reduced
    tensors live -- the rank's GPU under RCCL (on-device all-reduce), "cpu" under gloo."""
    hist = torch.as_tensor(hist, dtype=torch.int64).to(device).clone()
    cnt = torch.tensor([int(frames)], dtype=torch.int64, device=device)
    sec = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():  # also for a world of one: the collective path is the same code
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    return hist.cpu(), int(cnt.item()), float(sec.item())


def gather_floats(values, device="cpu"):
    """Every rank's list of floats, as [rank][i] (one all_gather of len(values) float64 at the end of a run; a single process
    gets [[...]]).  bench.py prints each rank's own ms per step with it, so that a straggler shows in the N > 1 line."""
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [mine.tolist()]
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [p.tolist() for p in parts]


def miou_from_hist(hist):
    """mIoU with the reference's epsilon (flow/base.py:332-336); hist = (intersection, |pred|, |target|)."""
    h = hist.double()
    union = h[1] + h[2] - h[0]
    return float((h[0] / (union + 1e-10)).mean())
