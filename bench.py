#!/usr/bin/env python3
"""Headline benchmark: segmentation FPS at 713x713 on BASELINE.json configs[1]
(PSPNet-ResNet50, key-frame + linear interpolation, frame_delta = 5), N = 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W        # N > 1 without a launcher: spawns the N ranks itself (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU -- it starts the N ranks as fresh
child processes (`python -m torch.distributed.run ... bench.py <same arguments>`, rendezvous on 127.0.0.1), lets rank 0's JSON
line through on stdout and exits with the children's return code.  Under N > 1 the step walks BASELINE configs[4]: 64 synthetic
713x713 clips sharded by clip (rank r owns clips r, r+N, ...; 4 key-frame windows per 21-frame clip, flow/dataset.py:64,112-114),
window after window, cyclically for the K timed steps; no data-path collective.

A step = one key-frame window of the hot path on each rank: FlowModel.predict(prev, next, ...) through
the C ABI (BOTH key frames segmented, exactly the work the reference does per predict call -- no cached
key frame) returning the fp32 logits of all 5 frames, per-frame argmax (emitted by the same fused tail that writes the logits),
uint8 masks copied to the host (the reference's timed region
"predict_interference", flow/base.py:269-277, at native 713x713 resolution).  Inputs are resident in
HBM when the clock starts.  value = frames all ranks produced / max-over-ranks wall time.
Everything is fp32 (the reference's precision); data and weights are synthetic (seeded).

`variants` (never the headline): the reference-exact 1072x1920 post-processing, the key-frame cache (one new key frame per
window), two windows in flight, the other BASELINE configs on one GPU (configs[0] per-frame PSPNet, configs[2] DeepLabv3-R101
+ logit warp, configs[3] ViT-S/16 + feature flow) and the reference's default real-video route (8 crops of a 1072x1920 frame).
"""
import argparse
import glob
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time

T_PROCESS = time.time()  # this process's start, as far as Python can tell (distributed.startup_s counts from the launcher's when there is one)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch and the package are imported by load_runtime(), AFTER main() has decided whether this process is a rank or only the
# launcher of the ranks: the launcher must never initialise a GPU (and has no use for a minute of imports).
torch = ops = shard = synth = FlowModel = KeyframeCache = FlowPredictor = FlowPSPNet = None


def load_runtime():
    global torch, ops, shard, synth, FlowModel, KeyframeCache, FlowPredictor, FlowPSPNet
    import torch as _torch
    from flood_uav_video_segmentation_amd import ops as _ops, shard as _shard, synth as _synth
    from flood_uav_video_segmentation_amd.flow.model import FlowModel as _FM, KeyframeCache as _KC
    from flood_uav_video_segmentation_amd.flow.predict import FlowPredictor as _FP
    from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet as _PSP
    torch, ops, shard, synth, FlowModel, KeyframeCache, FlowPredictor, FlowPSPNet = _torch, _ops, _shard, _synth, _FM, _KC, _FP, _PSP


SIZE = 713
N_DELTA = 5
CLASSES = 5
NUM_CLIPS = 64                   # BASELINE configs[4]: 64 synthetic clips sharded by clip
CLIP_FRAMES = 21                 # 4 key-frame windows of frame_delta 5 per clip (flow/dataset.py:64)
LONG_WINDOWS = 40                # variants.fps_keyframe_cache_lookahead_long_clip: a 201-frame clip
KEYFRAME_GFLOP = 727.44          # PSPNet-R50 encoder+decoder at 713^2
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16), no sparsity
NOMINAL_SCLK_MHZ = 2400          # MI355X_MICROARCH.md: the clock the peaks are quoted at
SPLIT_TERMS = 6                  # bf16 MFMA FLOPs executed per fp32 FLOP of the split-operand kernel (6 of the 9 cross products)
ARITHMETIC = ("fp32 tensors; in the implicit-GEMM kernels every fp32 operand is the exact sum of three bf16 terms (round-to-nearest residues) "
              "and six of the nine cross products (all of order <= 2^-16) run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three "
              "dropped ones are <= 2^-23 of a product.  Measured against float64 the kernel's error equals the fp32-MFMA kernel's "
              "(tests/test_gpu_ops.py::test_conv_igemm_split_operands, profiles/r03_split_operands.txt); FS_OPT_NO_SPLIT_BF16 selects the "
              "fp32-MFMA kernel (variants.fps_fp32_mfma_kernels_no_split)")


HIP_OPTS = {}  # development A/B only (--hip-opt): explicit fs_config options of the library, e.g. hip_no_res_touch, hip_no_fused_pool (unknown names are refused: model/hipnet.py::HIP_OPTIONS)


class HP:
    def __init__(self, layers=50):
        self.layers, self.classes, self.pretrained = layers, CLASSES, False
        for k, v in HIP_OPTS.items():
            setattr(self, k, v)


def build_id():
    """Identity of the kernels this process runs: sha256 over the library's sources (csrc/*.hip, *.h, Makefile) and the ABI
    header.  The .so is a pure function of them and of the image's hipcc, and unlike a git head it exists on the GPU box."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "flood_uav_video_segmentation_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")])
    for f in files + [os.path.join(ROOT, "include", "floodseg.h"), os.path.join(ROOT, "include", "floodseg_test.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def newest_pmc_summary():
    """(path, parsed json) of the newest profiles/rNN_pmc_traffic.json, or (None, None)."""
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        m = re.search(r"r(\d+)_pmc_traffic\.json$", f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None, None
    try:
        with open(best[1]) as fh:
            return best[1], json.load(fh)
    except (OSError, ValueError):
        return best[1], None


def kernel_symbol(label):
    """HIP-event label of the library's profile (split128x128, igemm128x64cat, ...) -> the kernel symbol rocprofv3 reports."""
    m = re.match(r"(igemm|split)(\d+)x(\d+)(cat)?$", label)
    split, bm, bn, cat = m.group(1) == "split", int(m.group(2)), int(m.group(3)), bool(m.group(4))
    # (split64x128: 2 x 2 waves like the 64 x 64 tile)
    waves = "4, 1" if split and bm == 128 else "2, 2"
    return f"conv_igemm_dma_f32<{bm}, {bn}, {waves}, {'true' if cat else 'false'}, {'true' if split else 'false'}>"


def pmc_traffic(dom_kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (tools/gpu_pmc_bench.sh; PMC counters cannot be read from inside the process).  Only a summary measured on THIS build
    (same build_id) is quoted; anything else gives traffic = null with the reason."""
    path, pmc = newest_pmc_summary()
    if path is None:
        return None, "no profiles/r*_pmc_traffic.json committed"
    rel = os.path.relpath(path, ROOT)
    if pmc is None:
        return None, f"{rel} unreadable"
    meta = pmc.get("meta")
    if not meta or "build_id" not in meta:
        return None, f"{rel} does not record the build it was measured on (a previous round's file): stale"
    if meta["build_id"] != build_id():
        return None, f"{rel} was measured on build {meta['build_id']} (git {meta.get('git_head', '?')}), this run is build {build_id()}: stale"
    hits = [k for k in pmc.get("kernels", {}) if kernel_symbol(dom_kernel) in k]
    if not hits or "hbm_bytes_per_launch" not in pmc["kernels"][hits[0]]:
        return None, f"{rel} has no FETCH_SIZE/WRITE_SIZE pair for {dom_kernel}"
    e = pmc["kernels"][hits[0]]
    extra = {k: (round(e[k], 4) if isinstance(e[k], float) else {a: round(b, 4) for a, b in e[k].items()})
             for k in ("mfma_utilisation", "lds_bank_conflict_share_of_lds_cycles", "wave_time_shares") if k in e}
    pmc_traffic.extra = extra  # SQ / GRBM counters of the same passes: the hardware's own MFMA-busy share, LDS conflicts, wave time split
    return round(e["hbm_bytes_per_launch"]), (
        f"(2*FETCH_SIZE + WRITE_SIZE)*1024 B averaged over the kernel's launches, {rel} (same build {meta['build_id']}, git {meta.get('git_head', '?')})")


pmc_traffic.extra = {}


class ClockSampler:
    """Shader clock and socket power of THIS card (hwmon: freq1_input, power1_input; read-only sysfs), sampled by a thread while the
    caller loops a step.  The card is found by its PCI address; `ok` is False when the sensors are not readable."""

    def __init__(self, torch, period_s=0.004):
        self.period_s, self.clk, self.pw, self.dir, self.addr = period_s, [], [], None, None
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            self.addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except AttributeError:
            return
        mons = [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
                if os.path.basename(os.path.realpath(os.path.join(d, "..", ".."))) == self.addr]
        if mons and self.rd("freq1_input", mons[0]) is not None and self.rd("power1_input", mons[0]) is not None:
            self.dir = mons[0]

    @property
    def ok(self):
        return self.dir is not None

    def rd(self, name, d=None):
        try:
            with open(os.path.join(d or self.dir, name)) as fh:
                return int(fh.read())
        except (OSError, ValueError, TypeError):
            return None

    def __enter__(self):
        import threading

        self.stop = threading.Event()

        def sampler():
            while not self.stop.is_set():
                f, w = self.rd("freq1_input"), self.rd("power1_input")
                if f is not None and w is not None:
                    self.clk.append(f / 1e6)
                    self.pw.append(w / 1e6)
                time.sleep(self.period_s)

        self.th = threading.Thread(target=sampler, daemon=True)
        if self.ok:
            self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.ok:
            self.th.join()
        return False


def clock_under_load(step, torch, windows=120, period_s=0.004):
    """Shader clock and socket power of THIS card while the headline step loops.  Run after the timed region.  None when the sensors
    are not readable."""
    cs = ClockSampler(torch, period_s)
    if not cs.ok:
        return None
    for i in range(20):  # the card may have dropped to its idle clock during the host-side work before this pass
        step(i)
    with cs:
        t0 = time.perf_counter()
        for i in range(windows):
            step(i)
        dt = time.perf_counter() - t0
    clk, pw = cs.clk, cs.pw
    if len(clk) < 8:
        return None
    return {"sclk_mhz_mean": round(sum(clk) / len(clk), 1), "sclk_mhz_min": round(min(clk), 1), "sclk_mhz_max": round(max(clk), 1),
            "power_w_mean": round(sum(pw) / len(pw), 1), "power_w_max": round(max(pw), 1), "power_cap_w": (cs.rd("power1_cap") or 0) / 1e6,
            "samples": len(clk), "windows": windows, "ms_per_step_during_sampling": round(dt / windows * 1e3, 4),
            "source": f"hwmon freq1_input / power1_input of {cs.addr}, one sample per {period_s * 1e3:.0f} ms over {windows} windows after the timed region; "
                      f"roofline.peak is quoted at the nominal {NOMINAL_SCLK_MHZ} MHz"}


STEADY_MIN_TIMED_S = 0.5   # a timed region shorter than this gets a steady-state block printed next to it
STEADY_BLOCK_S = 0.8       # ... of at least this long


def steady_state_block(step, torch):
    """A caller's `--steps K` may time well under a second (the driver: 20 windows = 0.08 s), and on this chip the clock a card
    holds depends on what ran in the last tens of milliseconds (profiles/r04_experiments.txt section 1).  So next to -- never instead
    of -- such a headline, the same step looped for >= 0.8 s right after the timed region, with per-step stamps and the card's
    shader clock sampled meanwhile: a box that had not reached its held clock in the timed region shows up as a gap between the two."""
    cs = ClockSampler(torch, 0.004)
    per = []
    torch.cuda.synchronize()
    with cs:
        t0 = tp = time.perf_counter()
        n = 0
        while n < 20 or (tp - t0 < STEADY_BLOCK_S and n < 200000):  # by the clock, not by an estimate: the block IS at least 0.8 s long
            step(n)
            n += 1
            tn = time.perf_counter()
            per.append(tn - tp)
            tp = tn
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"ms_per_step": round(dt / n * 1e3, 4), "median": round(median(per) * 1e3, 4), "steps": n, "seconds": round(dt, 3),
            "sclk_mhz_mean": round(sum(cs.clk) / len(cs.clk), 1) if len(cs.clk) >= 8 else None,
            "why": f"the timed region lasted under {STEADY_MIN_TIMED_S} s; this block ran right after it (same step, same inputs) and is NOT the headline"}


def timed(fn, steps, warmup, dev, per_step=None, marks=None):
    """K steps between barrier + synchronize brackets -> seconds.  per_step (a list): the wall time of every step is appended to it
    (perf_counter stamps between steps; every step function ends with its own stream synchronize, so a stamp is a completed step).
    marks (a dict): marks["first_timed_step"] = time.time() when the timed region starts."""
    for i in range(warmup):
        fn(i)
    shard.barrier(dev)
    torch.cuda.synchronize()
    if marks is not None:
        marks["first_timed_step"] = time.time()
    t0 = tp = time.perf_counter()
    for i in range(steps):
        fn(i)
        if per_step is not None:
            tn = time.perf_counter()
            per_step.append(tn - tp)
            tp = tn
    torch.cuda.synchronize()
    shard.barrier(dev)
    return time.perf_counter() - t0


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return 0.0 if n == 0 else xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def host_threads():
    """CPU threads this process may really use: cgroup quota, else affinity, capped at the 16-core share of a 1-GPU box."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(state, windows_cpu):
    """The oracle ("port" of the reference CPU PyTorch path) on the host cores: >= 4 windows of the same workload, ~10-25 s."""
    from oracle import flow_oracle, pspnet_oracle

    threads = host_threads()
    torch.set_num_threads(threads)
    enc = lambda x: pspnet_oracle.encoder(x, state, 50)  # noqa: E731
    dec = lambda f: pspnet_oracle.decoder(f, state)  # noqa: E731
    dl, dr = synth.dummy_grids(N_DELTA)
    with torch.no_grad():
        small = windows_cpu[0][0][:, :, :129, :129]
        flow_oracle.predict_segmentation(enc, dec, small, small, dl, dr, N_DELTA, True)  # warm the thread pool
        t0 = time.perf_counter()
        done = 0
        while (done < 4 or time.perf_counter() - t0 < 10.0) and time.perf_counter() - t0 < 25.0:
            prev, nxt = windows_cpu[done % len(windows_cpu)]
            out = flow_oracle.predict_segmentation(enc, dec, prev, nxt, dl, dr, N_DELTA, True)["pred"]
            out.max(1)[1].to(torch.uint8)
            done += 1
        dt = time.perf_counter() - t0
    model = None
    try:  # core count AND CPU model beside the baseline
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
    except OSError:
        pass
    return {"value": round(done * N_DELTA / dt, 4), "unit": "frames/s", "cores": threads, "cpu_model": model, "kind": "port",
            "sample": f"{done} windows ({dt:.1f} s) of the same config (PSPNet-R50, no_warp, n=5, 713x713, both key frames segmented per window) "
                      "through oracle/ on torch-CPU fp32"}


# what an RCCL run that could not exchange IPC handles leaves on stderr (hipIpcGetMemHandle / hipIpcOpenMemHandle failing under
# the wrong HSA_ENABLE_IPC_MODE_LEGACY, surfacing as an unhandled-HIP-error NCCL exception) ...
IPC_FAILURE = re.compile(r"hipIpc|IpcGetMemHandle|IpcOpenMemHandle|ncclUnhandledCudaError|ncclSystemError|NCCL error|RCCL error|unhandled (cuda|hip) error", re.I)
# ... and what a rank killed by a signal / a GPU fault leaves (torch.distributed.run reports the child's signal): never retried
SIGNAL_EXIT = re.compile(r"Signal \d+ \(SIG|exitcode\s*:\s*-\d+|core dumped|Memory access fault|HSA_STATUS_ERROR|Aborted", re.I)


def spawn_ranks(n):
    """`bench.py --gpus N` launched directly (no WORLD_SIZE): start the N ranks as FRESH child processes under
    torch.distributed.run and hand back their return code.  Nothing in this process has touched a GPU, and it does not
    replace itself (no exec): the children are ordinary subprocesses; rank 0's JSON line is relayed on stdout, the ranks' stderr
    on stderr.

    HSA_ENABLE_IPC_MODE_LEGACY: the image exports it as 0 here and on the GPU boxes (the host driver only supports dmabuf IPC;
    without it RCCL's hipIpcGetMemHandle fails with "invalid argument" -- the environment notes of this build pipeline), so the
    first attempt keeps the inherited value, 0 when unset.  It has never run under RCCL with N > 1 in a record of this repo, so
    the launcher does not bet the run on it -- but it only second-guesses THAT: the ranks are started once more (fresh children
    again, the opposite setting) only when they exited with a plain non-zero code BEFORE rank 0 printed its JSON line AND their
    stderr carries the IPC / RCCL-initialisation signature.  A Python exception, an out-of-memory kill, a bad argument, a signal or
    a GPU fault is handed back as it is (a faulting run must not get a silent second go).  Both attempts' return codes are printed
    on stderr, and the second attempt's JSON line carries the first one's (`distributed.launch.previous_attempt_rc`)."""
    import threading

    first = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    attempts = [first, "1" if first == "0" else "0"]
    rc, prev_rc = 1, None
    for attempt, ipc in enumerate(attempts):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = ipc
        env["FS_BENCH_LAUNCH_ATTEMPT"] = str(attempt)
        env["FS_BENCH_LAUNCHER_T0"] = repr(time.time())  # distributed.startup_s counts from here: spawn + imports + weights + warm-up
        if prev_rc is not None:
            env["FS_BENCH_PREV_ATTEMPT_RC"] = str(prev_rc)
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        err_tail = []

        def pump(stream=proc.stderr, keep=err_tail):
            for line in stream:
                sys.stderr.write(line)
                keep.append(line)
                del keep[:-400]
            sys.stderr.flush()

        th = threading.Thread(target=pump, daemon=True)
        th.start()
        printed = False
        for line in proc.stdout:
            printed = printed or line.startswith("{")
            sys.stdout.write(line)
            sys.stdout.flush()
        rc = proc.wait()
        th.join()
        if rc == 0 or printed or attempt + 1 == len(attempts):
            break
        text = "".join(err_tail)
        if rc < 0 or SIGNAL_EXIT.search(text) or not IPC_FAILURE.search(text):
            print(f"bench.py: the {n} ranks exited with code {rc} before a result line (HSA_ENABLE_IPC_MODE_LEGACY={ipc}); their stderr does not show "
                  "an IPC / RCCL-initialisation failure (or shows a signal / GPU fault): not relaunched", file=sys.stderr, flush=True)
            break
        prev_rc = rc
        print(f"bench.py: the {n} ranks exited with code {rc} before a result line with an IPC / RCCL-initialisation error on stderr "
              f"(HSA_ENABLE_IPC_MODE_LEGACY={ipc}); starting them once more with HSA_ENABLE_IPC_MODE_LEGACY={attempts[attempt + 1]}",
              file=sys.stderr, flush=True)
    if prev_rc is not None:
        print(f"bench.py: launch attempts returned {prev_rc} (HSA_ENABLE_IPC_MODE_LEGACY={attempts[0]}) then {rc} ({attempts[1]})", file=sys.stderr, flush=True)
    return rc


def launch_check(args):
    """--launch-check: what an N-rank run does around its timed loop, on CPU over gloo and with no model: rendezvous, the rank's
    shard of the 64-clip schedule, barrier, the end-of-run all_reduce / all_gather.  Prints a line with NO metric / value."""
    import torch as _torch
    from flood_uav_video_segmentation_amd import shard as _shard
    rank, _, world = _shard.init("gloo")
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    schedule = _shard.clip_window_schedule(NUM_CLIPS if world > 1 else 1, CLIP_FRAMES, N_DELTA, rank, world)
    _shard.barrier()
    _, frames, sec = _shard.reduce_run(_torch.zeros(3, CLASSES, dtype=_torch.int64), len(schedule) * N_DELTA, 1.0 + rank)
    per_rank = _shard.gather_floats([float(len(schedule))])
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "frames_of_all_shards": frames, "max_seconds": sec,
                          "windows_per_rank": [int(r[0]) for r in per_rank], "distributed": dict(zip(("backend", "world_size"), _shard.describe())),
                          "launch_attempt": int(os.environ.get("FS_BENCH_LAUNCH_ATTEMPT", 0))}), flush=True)
    if world > 1:
        _torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary variants")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and the reduction runs over gloo")
    ap.add_argument("--lib", default=None,
                    help="development A/B only: load this build of the library instead of the in-tree libfloodseg.so (recorded in the JSON line)")
    ap.add_argument("--hip-opt", action="append", default=[], metavar="NAME[=INT]",
                    help="development A/B only: set this explicit library option (hparams attribute, e.g. hip_no_res_touch; unknown names are refused) on every network "
                         "the run builds; recorded in the JSON line")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU, no measurement: the N ranks rendezvous over gloo, build their shard of the configs[4] schedule, run the "
                         "end-of-run collectives and rank 0 prints a line WITHOUT a metric (rehearses the N-rank launch on CPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # launcher only: no torch, no GPU in this process
    if args.launch_check:
        return launch_check(args)
    for o in args.hip_opt:
        name, _, val = o.partition("=")
        HIP_OPTS[name] = int(val) if val else True
    if args.lib:
        from flood_uav_video_segmentation_amd import _lib as _l
        _l.LIB_PATH = os.path.abspath(args.lib)
        _l.ALLOW_MISSING = True  # an older build may lack the newest entry points; the headline step does not use them
    load_runtime()

    rank, local_rank, world = shard.init("gloo" if args.rehearse_on_one_gpu else None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_grad_enabled(False)
    backend, dist_world = shard.describe()
    rdev = "cpu" if backend != "nccl" else dev  # RCCL reduces on-device tensors; gloo (rehearsal / single process) on the host

    # ---- model: weights replicated on every rank (regenerated from the same seed, no broadcast needed)
    state = synth.make_pspnet_state(50, CLASSES, seed=0)
    net = FlowPSPNet(HP()).eval()
    net.load_state_dict(state)
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()

    # ---- data, resident in HBM.  N = 1: BASELINE configs[1], the 4 key-frame windows of ONE clip (clip 0).  N > 1: BASELINE
    # configs[4], this rank's shard of the 64 clips (clip c = seed 1000 + c; rank r owns c = r, r+N, ...), 4 windows each.
    schedule = shard.clip_window_schedule(NUM_CLIPS if world > 1 else 1, CLIP_FRAMES, N_DELTA, rank, world)
    shard_windows = len(schedule)
    # only the windows the run will visit are generated and kept resident (step i takes window i mod len): a short run on a big
    # shard does not spend a minute synthesising clips it never touches
    schedule = schedule[:max(4, min(shard_windows, max(args.steps, args.warmup, 5)))]
    keys_of = {}
    for c in sorted({c for c, _, _ in schedule}):
        keys_of[c] = synth.make_clip(CLIP_FRAMES, SIZE, seed=1000 + c, only=list(range(0, CLIP_FRAMES, N_DELTA)))
    first = schedule[0][0]
    windows_cpu = [(keys_of[first][i:i + 1], keys_of[first][i + 1:i + 2]) for i in range(4)] if rank == 0 else []
    keys_dev = {c: k.to(dev) for c, k in keys_of.items()}
    windows = [(keys_dev[c][k0 // N_DELTA:k0 // N_DELTA + 1], keys_dev[c][k1 // N_DELTA:k1 // N_DELTA + 1]) for c, k0, k1 in schedule]
    del keys_of
    nwin = len(windows)
    dl, dr = [[g.to(dev) for g in gs] for gs in synth.dummy_grids(N_DELTA)]
    host_masks = torch.empty((N_DELTA, SIZE, SIZE), dtype=torch.uint8).pin_memory()
    host_post = torch.empty((N_DELTA, 1072, 1920), dtype=torch.uint8).pin_memory()
    net.reserve(2, SIZE, SIZE)  # the library's workspace for this geometry, once, before the clock starts (fs_reserve)

    def step_native(i):
        prev, nxt = windows[i % nwin]
        out = fm.predict(prev, nxt, dl, dr, N_DELTA, None, with_mask=True)  # logits [5,K,713,713] AND their argmax, from one fused tail
        host_masks.copy_(out["mask"], non_blocking=True)
        torch.cuda.current_stream().synchronize()  # masks are on the host when the step ends

    def enqueue_native(i):
        """step_native without the wait: what the HOST does per window (Python, ctypes, launches, the async copy)."""
        prev, nxt = windows[i % nwin]
        host_masks.copy_(fm.predict(prev, nxt, dl, dr, N_DELTA, None, with_mask=True)["mask"], non_blocking=True)

    step_s, marks = [], {}
    cs_timed = ClockSampler(torch, 0.01)  # this rank's card while the timed region runs (one sysfs read per 10 ms)
    with cs_timed:
        elapsed = timed(step_native, args.steps, args.warmup, dev, step_s, marks)
    _, frames_total, elapsed_max = shard.reduce_run(torch.zeros(3, CLASSES, dtype=torch.int64), args.steps * N_DELTA, elapsed, rdev)
    fps = frames_total / elapsed_max
    # process start (the launcher's, when bench.py spawned the ranks itself) -> first timed step, per rank: imports, weights, data,
    # workspace, warm-up and the barrier -- what a driver-side timeout on a many-GPU node would have been spent on
    t_ref = float(os.environ.get("FS_BENCH_LAUNCHER_T0", T_PROCESS))
    # every rank's own figures (mean and median step, start-up), so that a straggler is visible in the N > 1 line
    # a short timed region says little about the clock the card settles at: the same step for >= 0.8 s right behind it (every rank
    # runs it, so the ranks stay symmetric; rank 0 reports its own)
    steady = steady_state_block(step_native, torch) if elapsed < STEADY_MIN_TIMED_S else None
    # host time to ENQUEUE one window (no wait; a short burst, so the launch queue never pushes back): with one process per GPU on a
    # many-GPU node this is what the ranks' share of the host cores has to sustain -- a rank is host-bound once it reaches ms_per_step
    torch.cuda.synchronize()
    t_enq = time.perf_counter()
    for i in range(4):
        enqueue_native(i)
    enqueue_ms = (time.perf_counter() - t_enq) / 4 * 1e3
    torch.cuda.synchronize()
    # the clock this rank's card held: sampled inside the timed region when that was long enough for a few samples, else in the block above
    sclk = (sum(cs_timed.clk) / len(cs_timed.clk)) if len(cs_timed.clk) >= 4 else ((steady or {}).get("sclk_mhz_mean") or -1.0)
    per_rank = shard.gather_floats([elapsed / args.steps * 1e3, median(step_s) * 1e3, marks["first_timed_step"] - t_ref, enqueue_ms, float(sclk),
                                    float(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))], rdev)

    result = {
        "metric": "segmentation FPS @713x713 (PSPNet-ResNet50 keyframe + linear interp, frame_delta=5)",
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
        # median window latency (the reference reports the mean of its per-call timer, flow/base.py:321-328): per-step
        # perf_counter stamps inside the same timed region, this rank's steps (N > 1: the slowest rank's median)
        "median_ms_per_step": round(max(r[1] for r in per_rank), 4),
        "steady_state": steady,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "arithmetic": ARITHMETIC, "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: PSPNet-ResNet50 keyframe + linear interp (no_warp=True, feature_based=False), "
                                "frame_delta=5, 713x713, one window per step per GPU, both key frames segmented per window, "
                                "argmax uint8 masks copied to host") if world == 1 else
                               (f"BASELINE configs[4]: {NUM_CLIPS} synthetic 713x713 clips sharded by clip over {world} GPUs (rank r owns clips "
                                f"r, r+{world}, ...: {shard_windows // 4} clips x 4 key-frame windows on this rank), PSPNet-ResNet50 keyframe + linear "
                                "interp, frame_delta=5; one window per step per GPU, the timed steps walk the rank's shard window after "
                                "window (cyclically), both key frames segmented per window, argmax uint8 masks copied to host"),
                   "frames_per_step_per_gpu": N_DELTA, "clips_total": NUM_CLIPS if world > 1 else 1,
                   "windows_in_this_ranks_shard": shard_windows, "windows_resident_and_visited": nwin,
                   "parallelism": f"{world} independent clip shard(s), no data-path collective"},
        # what the launcher really set up: "nccl" IS RCCL on ROCm; a single process has no process group
        "distributed": {"backend": backend, "rccl_world_size": dist_world if backend == "nccl" else 0, "world_size": dist_world,
                        "rank_ms_per_step": {"min": round(min(r[0] for r in per_rank), 4), "max": round(max(r[0] for r in per_rank), 4),
                                             "per_rank": [round(r[0], 4) for r in per_rank]},
                        # what an N = 8 line is read with: the host side of every rank -- time to enqueue one window
                        # (host-bound when it approaches ms_per_step), the cores the rank may run on, and the clock its card held
                        "host_enqueue_ms_per_step": {"max": round(max(r[3] for r in per_rank), 4), "per_rank": [round(r[3], 4) for r in per_rank],
                                                     "what": "host wall time to enqueue one window without waiting (4 windows, after the timed region)"},
                        "host_cpus": {"cpu_count": os.cpu_count(), "affinity_per_rank": [int(r[5]) for r in per_rank], "ranks": world,
                                      "cpus_per_rank": round(min(r[5] for r in per_rank) / max(world, 1), 2)},
                        "sclk_mhz_mean": {"per_rank": [round(r[4], 1) if r[4] > 0 else None for r in per_rank],
                                          "source": "hwmon freq1_input of each rank's card during its timed region (the steady-state block when that was too short)"},
                        "startup_s": {"max": round(max(r[2] for r in per_rank), 2), "per_rank": [round(r[2], 2) for r in per_rank],
                                      "from": "bench.py launcher start" if "FS_BENCH_LAUNCHER_T0" in os.environ else "this process's start",
                                      "to": "first timed step (after imports, weights, resident inputs, fs_reserve, warm-up, barrier)"},
                        "launch": {"by": "bench.py spawn_ranks" if "FS_BENCH_LAUNCH_ATTEMPT" in os.environ else ("torchrun" if world > 1 else "single process"),
                                   "attempt": int(os.environ.get("FS_BENCH_LAUNCH_ATTEMPT", 0)),
                                   "previous_attempt_rc": int(os.environ["FS_BENCH_PREV_ATTEMPT_RC"]) if "FS_BENCH_PREV_ATTEMPT_RC" in os.environ else None,
                                   "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
                        "collectives": "end-of-run all_reduce of int64 frame count + float64 seconds (and int64[3,K] histograms in tools/predict_video.py); "
                                       "none inside the timed loop"},
        "build_id": build_id() if not args.lib else f"--lib {os.path.basename(args.lib)} (not the in-tree build)",
        "hip_options": dict(HIP_OPTS) or None,  # null = the shipped defaults (what the driver runs)
        "reference_claim_fps_other_hw": 76.85,
    }

    if rank == 0:
        # ---- parity on the bench inputs: HIP masks vs the reference's own masks (golden fixture, clip seed 1000)
        try:
            import numpy as np
            with np.load(os.path.join(ROOT, "tests", "golden", "predict_713.npz")) as z:
                ref_mask = torch.from_numpy(z["cfg2_mask"]).to(dev)
            # the fixture's window is key frames (0, 5) of clip 1000 = windows[0] of rank 0
            got = fm.predict_masks(windows[0][0], windows[0][1], dl, dr, N_DELTA)
            hist = ops.iou_hist(got, ref_mask, CLASSES)
            result["parity"] = {"mask_agreement_vs_reference": round((got == ref_mask).float().mean().item(), 6),
                                "miou_vs_reference_masks": round(shard.miou_from_hist(hist.cpu()), 6),
                                "miou_delta_pp": round((1.0 - shard.miou_from_hist(hist.cpu())) * 100, 4)}
        except Exception as e:  # noqa: BLE001
            result["parity"] = {"error": repr(e)[:200]}

    # ---- roofline of the dominant kernel: HIP events around every launch, same steps repeated right after the timed region
    net._hip_net.profile(True)
    prof_steps = max(1, min(args.steps, 5))
    for i in range(prof_steps):
        step_native(i)
    rows = net._hip_net.profile_dump()
    net._hip_net.profile(False)
    if rank == 0:
        per = {}
        for name, kernel, flops, nbytes, ms in rows:
            d = per.setdefault(kernel, {"ms": 0.0, "flops": 0.0, "launches": 0})
            d["ms"] += ms
            d["flops"] += flops
            d["launches"] += 1
        conv = {k: v for k, v in per.items() if k.startswith(("igemm", "split"))}  # the implicit-GEMM launches (direct convs, Winograd position GEMMs, Linears)
        dom_name, dom = max(conv.items(), key=lambda kv: kv[1]["ms"])
        is_split = dom_name.startswith("split")
        # split-operand kernel: the matrix cores execute 6 bf16 MFMA FLOPs per algorithmic (fp32) FLOP; the roofline is the bf16 pipe's
        alg = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        ach = alg * (SPLIT_TERMS if is_split else 1)
        peak = PEAK_BF16_MFMA_TFLOPS if is_split else PEAK_F32_MFMA_TFLOPS
        alg_bytes = sum(r[3] for r in rows if r[1] == dom_name) / max(1, dom["launches"])
        traffic, traffic_note = pmc_traffic(dom_name)
        all_ms = sum(v["ms"] for v in conv.values())
        all_fl = sum(v["flops"] for v in conv.values())
        result["roofline"] = {
            "bound": "mfma", "kernel": kernel_symbol(dom_name), "achieved": round(ach, 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
            "pipe": "bf16 MFMA (v_mfma_f32_32x32x16_bf16), 6 executed FLOPs per algorithmic fp32 FLOP" if is_split else "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
            "algorithmic_fp32_tflops": round(alg, 2),
            "traffic_unit": "B per launch", "traffic_source": traffic_note, "algorithmic_bytes_per_launch": round(alg_bytes),
            "avg_launch_ms": round(dom["ms"] / dom["launches"], 5), "launches_per_step": dom["launches"] // prof_steps,
            "timing_source": (f"HIP events on the library's stream around every launch of {prof_steps} profiled repeats of the same step, run right "
                              "AFTER the timed region (the events cost a few % themselves: the per-kernel figures sum to slightly more than "
                              "ms_per_step, which comes from the un-instrumented timed region)"),
            "gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
            "pmc": pmc_traffic.extra,
            "all_conv_kernels": {"algorithmic_fp32_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2), "ms_per_step": round(all_ms / prof_steps, 4),
                                 "gflop_per_step": round(all_fl / prof_steps / 1e9, 2)},
            "per_kernel_ms_per_step": {k: round(v["ms"] / prof_steps, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if "wino_fused" in per:
            # the one-kernel Winograd F(4x4,3x3) launches (deep stem, conv2 of layer1 / layer2): FLOPs EXECUTED on the matrix cores
            # (36 products per 16 outputs) and the direct-conv rate they stand for (x4) -- the latter is not a roofline fraction
            wf = per["wino_fused"]
            wf_tf = wf["flops"] / (wf["ms"] * 1e-3) / 1e12
            result["roofline"]["fused_winograd_kernels"] = {
                "achieved_executed": round(wf_tf, 2), "frac_of_mfma_peak": round(wf_tf / PEAK_F32_MFMA_TFLOPS, 4), "ms_per_step": round(wf["ms"] / prof_steps, 4),
                "launches_per_step": wf["launches"] // prof_steps, "gflop_executed_per_step": round(wf["flops"] / prof_steps / 1e9, 2),
                "direct_conv_equivalent_tflops": round(4 * wf_tf, 1)}
            _, pmc = newest_pmc_summary()
            if pmc and pmc.get("meta", {}).get("build_id") == build_id():  # the same --pmc passes, same build only
                for kname, e in pmc.get("kernels", {}).items():
                    if "wino4_" in kname:
                        tag = "warp_specialised" if "wino4_ws" in kname else "two_workgroups_per_cu"
                        result["roofline"]["fused_winograd_kernels"]["pmc_" + tag] = {
                            k: (round(e[k], 4) if isinstance(e[k], float) else {a: round(b, 4) for a, b in e[k].items()})
                            for k in ("mfma_utilisation", "l2_hit_rate", "lds_bank_conflict_share_of_lds_cycles", "hbm_bytes_per_launch", "wave_time_shares") if k in e}

    if not args.no_extras and world == 1 and "roofline" in result:
        try:
            cp = result["roofline"]["clock_and_power_under_load"] = clock_under_load(step_native, torch)
            if cp:  # the same achieved rate against the peak at the clock the card holds over the window (frac stays at the nominal clock)
                result["roofline"]["frac_of_peak_at_held_clock"] = round(result["roofline"]["frac"] * NOMINAL_SCLK_MHZ / cp["sclk_mhz_mean"], 4)
        except Exception as e:  # noqa: BLE001  (a sensor that cannot be read must not cost the bench line)
            result["roofline"]["clock_and_power_under_load"] = {"error": repr(e)[:200]}
    if not args.no_extras and world == 1:  # the variants are single-GPU figures: measured by the N = 1 run only
        result["variants"] = variants(args, dev, rdev, net, state, fm, windows, dl, dr, host_masks, host_post)

    if rank == 0:
        # direct-convolution-equivalent rate (727.44 GFLOP per key frame); the heavy 3x3 convs run as Winograd
        # F(6x6,3x3) and the head skips the pyramid channels, which together execute ~2.3x fewer FLOPs, so this "effective"
        # figure may exceed the fp32 MFMA peak -- it is NOT a roofline fraction
        result["effective_tflops_direct_conv_equivalent_per_gpu"] = round(2 * KEYFRAME_GFLOP * 1e-3 * (fps / world) / N_DELTA, 2)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(state, windows_cpu)
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def variants(args, dev, rdev, net, state, fm, windows, dl, dr, host_masks, host_post):
    """Secondary, driver-timed figures (each: same barrier + synchronize bracket, max over ranks).  FPS = output frames / s."""
    out = {}
    steps = args.steps
    few = max(1, min(steps, 40))  # the heavier variants: bounded so that the default run stays within minutes

    def run(name, fn, nsteps, frames_per_step, warm=2):
        e = timed(fn, nsteps, warm, dev)
        _, f, e = shard.reduce_run(torch.zeros(1, dtype=torch.int64), nsteps * frames_per_step, e, rdev)
        out[name] = round(f / e, 3)

    # (ii) the reference-exact post-processing: bilinear upsample to 1072x1920 + argmax (flow/base.py:275-277), fused
    def step_post(i):
        prev, nxt = windows[i % 4]
        logits = fm.predict(prev, nxt, dl, dr, N_DELTA, None)["pred"]
        host_post.copy_(ops.resize_argmax_u8(logits, (1072, 1920)), non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_post1072x1920_reference_exact_timed_region", step_post, steps, N_DELTA)

    # (iii) key-frame cache (product feature: FlowModel.predict(..., key_cache=...)): consecutive windows of a clip share a
    # key frame, so ONE frame is segmented per window; bit-identical masks (tests/test_gpu_fullsize.py)
    cache = KeyframeCache()

    def step_cached(i):
        w = i % 4
        if w == 0:
            cache.clear()  # a new clip starts: its first window segments both key frames
        prev, nxt = windows[w]
        mask = fm.predict_masks(prev, nxt, dl, dr, N_DELTA, None, key_cache=cache.window(5 * w, 5 * w + 5))
        host_masks.copy_(mask, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_keyframe_cache_one_new_keyframe_per_window", step_cached, steps, N_DELTA)

    # (iii-b) the same with one window of look-ahead (FlowPredictor.predict_clip): the new key frames of TWO consecutive windows
    # go through the network as one batch of two, so the cached run also keeps the efficiency of a full batch.  One step = the
    # four windows of a 21-frame clip (20 frames), masks of every window copied to the host.
    clip_items = [{"frame_prev": windows[w][0], "frame_next": windows[w][1], "mvs_left": dl, "mvs_right": dr, "key_ids": (5 * w, 5 * w + 5)}
                  for w in range(4)]
    pclip = FlowPredictor(fm, CLASSES, (SIZE, SIZE), compute_metrics=False)
    host_clip = torch.empty((4 * N_DELTA, SIZE, SIZE), dtype=torch.uint8).pin_memory()

    def step_clip(i):
        for w, masks in enumerate(pclip.predict_clip(clip_items, to_host=False)):
            host_clip[w * N_DELTA:(w + 1) * N_DELTA].copy_(masks, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_keyframe_cache_with_one_window_lookahead_clip_of_4_windows", step_clip, max(1, steps // 4), 4 * N_DELTA)

    # (iii-c) the streaming product mode on a LONG clip (the figure above is set by "5 key frames serve 4 windows"): LONG_WINDOWS
    # consecutive windows = LONG_WINDOWS + 1 key frames, each segmented exactly once, two per network pass.  The key frames are the
    # five resident ones walked back and forth (0 1 2 3 4 3 2 1 0 ...: consecutive keys always differ; the work per pass does not
    # depend on the pixels) under running frame ids, so the cache sees a 201-frame video.  One step = the whole clip.
    keys5 = [windows[0][0]] + [windows[w][1] for w in range(4)]

    def key_of(j):
        j %= 8
        return keys5[j if j <= 4 else 8 - j]
    long_items = [{"frame_prev": key_of(w), "frame_next": key_of(w + 1), "mvs_left": dl, "mvs_right": dr, "key_ids": (N_DELTA * w, N_DELTA * w + N_DELTA)}
                  for w in range(LONG_WINDOWS)]
    host_ring = torch.empty((4 * N_DELTA, SIZE, SIZE), dtype=torch.uint8).pin_memory()
    long_steps = max(4, steps // 20)

    def step_long(i):
        for w, masks in enumerate(pclip.predict_clip(long_items, to_host=False)):
            host_ring[(w % 4) * N_DELTA:(w % 4 + 1) * N_DELTA].copy_(masks, non_blocking=True)  # every window's masks go to the host
        torch.cuda.current_stream().synchronize()
    run("fps_keyframe_cache_lookahead_long_clip", step_long, long_steps, LONG_WINDOWS * N_DELTA, warm=1)

    # ... and with FOUR new key frames per network pass (predict_clip(keys_per_pass=4): three windows of look-ahead)
    def step_long4(i):
        for w, masks in enumerate(pclip.predict_clip(long_items, to_host=False, keys_per_pass=4)):
            host_ring[(w % 4) * N_DELTA:(w % 4 + 1) * N_DELTA].copy_(masks, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_keyframe_cache_lookahead_long_clip_4_keys_per_pass", step_long4, long_steps, LONG_WINDOWS * N_DELTA, warm=1)
    out["long_clip"] = {"windows": LONG_WINDOWS, "key_frames_segmented": LONG_WINDOWS + 1, "frames": LONG_WINDOWS * N_DELTA, "clips_timed": long_steps,
                        "note": "FlowPredictor.predict_clip: key-frame cache + one window of look-ahead; masks of every window copied to the host; "
                                "bit-identical to the uncached windows (tests/test_gpu_fullsize.py)"}

    # (iv) two windows in flight: a second library handle (own workspace) on a second stream, two windows per step
    net2 = FlowPSPNet(HP()).eval()
    net2.load_state_dict(state)
    fm2 = FlowModel(net2, feature_based=False, no_warp=True).eval()
    side = torch.cuda.Stream()
    host_masks2 = torch.empty((N_DELTA, SIZE, SIZE), dtype=torch.uint8).pin_memory()

    # (iv-a) the long clip with TWO clips in flight: a second predictor (second handle, second stream) walks its own copy of the clip,
    # window by window in step with the first
    pclip2 = FlowPredictor(fm2, CLASSES, (SIZE, SIZE), compute_metrics=False)
    host_ring2 = torch.empty((4 * N_DELTA, SIZE, SIZE), dtype=torch.uint8).pin_memory()

    def step_long_two(i):
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        g1, g2 = pclip.predict_clip(long_items, to_host=False), pclip2.predict_clip(long_items, to_host=False)
        for w in range(LONG_WINDOWS):
            host_ring[(w % 4) * N_DELTA:(w % 4 + 1) * N_DELTA].copy_(next(g1), non_blocking=True)
            with torch.cuda.stream(side):
                host_ring2[(w % 4) * N_DELTA:(w % 4 + 1) * N_DELTA].copy_(next(g2), non_blocking=True)
        for g in (g1, g2):
            assert next(g, None) is None
        main_s.synchronize()
        side.synchronize()
    run("fps_keyframe_cache_lookahead_long_clip_two_clips_in_flight", step_long_two, max(2, long_steps // 2), 2 * LONG_WINDOWS * N_DELTA, warm=1)
    del pclip2

    def step_two(i):
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        prev, nxt = windows[(2 * i) % 4]
        host_masks.copy_(ops.argmax_u8(fm.predict(prev, nxt, dl, dr, N_DELTA, None)["pred"]), non_blocking=True)
        with torch.cuda.stream(side):
            prev2, nxt2 = windows[(2 * i + 1) % 4]
            host_masks2.copy_(ops.argmax_u8(fm2.predict(prev2, nxt2, dl, dr, N_DELTA, None)["pred"]), non_blocking=True)
        main_s.synchronize()
        side.synchronize()
    run("fps_two_windows_in_flight_two_streams", step_two, max(1, steps // 2), 2 * N_DELTA)
    del net2, fm2

    # (iv-b) the headline step with the implicit GEMMs on the fp32-MFMA kernel (FS_OPT_NO_SPLIT_BF16): the round-2 arithmetic
    class HP32(HP):
        hip_no_split_bf16 = True
    net32 = FlowPSPNet(HP32()).eval()
    net32.load_state_dict(state)
    fm32 = FlowModel(net32, feature_based=False, no_warp=True).eval()

    def step_f32(i):
        prev, nxt = windows[i % 4]
        host_masks.copy_(fm32.predict(prev, nxt, dl, dr, N_DELTA, None, with_mask=True)["mask"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_fp32_mfma_kernels_no_split", step_f32, steps, N_DELTA)
    del net32, fm32

    # (v) BASELINE configs[0] on the GPU: single-frame PSPNet inference over a 4-frame clip, one frame per step
    frames4 = [windows[i][0] for i in range(4)]

    def step_single(i):
        lo = net.segment(frames4[i % 4])
        _, mask = ops.seg_tail(lo, None, [], [], 1, (SIZE, SIZE), True, want_logits=False, want_mask=True)
        host_masks[:1].copy_(mask, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_config0_pspnet_r50_single_frame", step_single, steps, 1)

    # (vi) the reference's default real-video route (no_cropping=False): 1072x1920 frames, 8 overlapping 713x713 crops of both
    # key frames batched through the network, fused tail + softmax + float64 canvas, masks at 1072x1920 (flow/base.py:182-209)
    hd = synth.make_clip(16, (1072, 1920), seed=1200, only=[0, 5, 10, 15]).to(dev)
    gl, gr = [[g.to(dev) for g in gs] for gs in synth.make_grids(N_DELTA, 67, 120, seed=2100, frame=(1072, 1920), jitter=0.01)]
    fmw = FlowModel(net, feature_based=False, no_warp=False).eval()
    pred = FlowPredictor(fmw, CLASSES, (1072, 1920), crop=(SIZE, SIZE), compute_metrics=False)

    def step_crops(i):
        w = i % 3
        host_post.copy_(pred.predict_window(hd[w:w + 1], hd[w + 1:w + 2], gl, gr, to_host=False), non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_real_video_route_8crops_1072x1920_warp", step_crops, max(1, few // 2), N_DELTA, warm=1)

    # ... and as tools/predict_video.py runs it by default: with the key-frame cache (8 new crop inferences per window, not 16)
    predc = FlowPredictor(fmw, CLASSES, (1072, 1920), crop=(SIZE, SIZE), compute_metrics=False, cache_keyframes=True)

    def step_crops_cached(i):
        w = i % 3
        if w == 0:
            predc.key_cache.clear()  # a new clip: its first window segments both key frames
        host_post.copy_(predc.predict_window(hd[w:w + 1], hd[w + 1:w + 2], gl, gr, to_host=False, key_ids=(5 * w, 5 * w + 5)), non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_real_video_route_8crops_keyframe_cache", step_crops_cached, max(3, few // 2), N_DELTA, warm=1)
    del pred, predc, hd

    # (vii) BASELINE configs[2]: DeepLabv3-ResNet101 key frames + optical-flow warp of the logits
    from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3
    wl, wr = [[g.to(dev) for g in gs] for gs in synth.make_grids(N_DELTA, 44, 44, seed=2000)]
    dl3 = FlowDeepLabv3(HP(101)).eval()
    dl3.load_state_dict(synth.make_deeplab_state(101, CLASSES, 0))
    fm3 = FlowModel(dl3, feature_based=False, no_warp=False).eval()

    def step_cfg2(i):
        prev, nxt = windows[i % 4]
        host_masks.copy_(fm3.predict_masks(prev, nxt, wl, wr, N_DELTA), non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_config2_deeplabv3_r101_logit_warp", step_cfg2, few, N_DELTA)
    del dl3, fm3

    # (viii) BASELINE configs[3]: Segmenter ViT-S/16 key frames + feature propagation (our extension: the reference has none)
    from flood_uav_video_segmentation_amd.model.vit import VITSegmentModel
    vit = VITSegmentModel(CLASSES, 704, patch_size=16, d_model=384, n_layers=12, dec_layers=2, **HIP_OPTS).eval()
    vit.load_state_dict(synth.make_vit_state(CLASSES, 704, 16, 384, 12, 2, seed=0))
    fmv = FlowModel(vit, feature_based=True, no_warp=False).eval()

    def step_cfg3(i):
        prev, nxt = windows[i % 4]
        host_masks.copy_(fmv.predict(prev, nxt, wl, wr, N_DELTA, None, with_mask=True)["mask"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    run("fps_config3_vit_s16_feature_flow", step_cfg3, few, N_DELTA)
    return out


if __name__ == "__main__":
    main()
