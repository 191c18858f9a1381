"""Grid producer on the HIP path -- the step right before the hot path:
dataset/flow/extract_motion_vectors.py:21-43 turns the H.264 16x16 block motion vectors of one frame into the
forward (`grids/`) and inverse (`inv_grids/`) sampling grids FlowData loads (flow/dataset.py:138-146, 239-240).
Decoding the video itself (mvextractor / ffmpeg) stays out of scope.
"""
import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr

BLOCK = 16
HEIGHT, WIDTH = 1072, 1920  # the geometry get_default_grid() is built for (flow/model.py:11)


def motion_vectors_to_grids(motion_vectors, frame_h, frame_w):
    """motion_vectors: int tensor/array [N, >=7] rows (source, w, h, src_x, src_y, dst_x, dst_y, ...).
    Returns (grid, inv_grid): float64 CUDA tensors [67, 120, 2], last dim (x, y) in [-1, 1]."""
    lib = _lib.load()
    mv = torch.as_tensor(np.asarray(motion_vectors) if not isinstance(motion_vectors, torch.Tensor) else motion_vectors)
    if mv.numel() and (mv.dim() != 2 or mv.shape[1] < 7):
        raise RuntimeError(f"motion vectors must be [N, >=7], got {tuple(mv.shape)}")
    if mv.numel():
        if not bool((mv[:, 0] == -1).all()):
            raise AssertionError("only past-frame references (source == -1) are supported")       # :27
        if not bool(((mv[:, 1] == BLOCK) & (mv[:, 2] == BLOCK)).all()):
            raise AssertionError("only 16x16 macroblocks are supported")                         # :31
    mv = mv.to(device="cuda", dtype=torch.int32).contiguous()
    hb, wb = HEIGHT // BLOCK, WIDTH // BLOCK
    grid = torch.empty((hb, wb, 2), dtype=torch.float64, device="cuda")
    inv = torch.empty_like(grid)
    owners = torch.empty(2 * hb * wb, dtype=torch.int32, device="cuda")
    n = mv.shape[0] if mv.numel() else 0
    stride = mv.shape[1] if mv.numel() else 7
    check(lib.fs_mv_to_grids(ptr(mv) if n else None, n, stride, hb, wb, BLOCK, int(frame_h), int(frame_w), ptr(owners), ptr(grid),
                             ptr(inv), stream_ptr()))
    return grid, inv


def save_grid(path, grid):
    """On-disk format of the reference: float64 (67,120,2) .npy (extract_motion_vectors.py:101-104)."""
    np.save(path, grid.detach().cpu().numpy().astype(np.float64))


def load_grid(path):
    """FlowData._load_grid (flow/dataset.py:239-240): float64 .npy -> float32 tensor."""
    return torch.from_numpy(np.load(path, allow_pickle=False)).float()
