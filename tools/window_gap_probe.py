#!/usr/bin/env python3
"""Where does the idle time between two windows of bench.py's step go?  (The kernel trace shows ~0.2 ms between the end of the mask
copy of window i and the first kernel of window i + 1.)  Host-side timestamps of one step: start -> entry of the library's forward
call -> its return -> predict() returned -> copy enqueued -> wait over, for three ways of waiting.  Needs an MI355X."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib, synth  # noqa: E402
from flood_uav_video_segmentation_amd.flow.model import FlowModel  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402
from tools.bench_configs import HP  # noqa: E402

N = 5


def main():
    dev = "cuda"
    psp = FlowPSPNet(HP(50)).eval()
    psp.load_state_dict(synth.make_pspnet_state(50, 5, 0))
    fm = FlowModel(psp, feature_based=False, no_warp=True).eval()
    keys = synth.make_clip(21, 713, seed=1000, only=[0, 5, 10, 15, 20]).to(dev)
    dl, dr = [[g.to(dev) for g in gs] for gs in synth.dummy_grids(N)]
    host = torch.empty((N, 713, 713), dtype=torch.uint8).pin_memory()
    lib = _lib.load()
    marks = {}
    inner = lib.fs_segment_forward2

    def wrapped(*a):
        marks["enter"] = time.perf_counter()
        r = inner(*a)
        marks["leave"] = time.perf_counter()
        return r
    lib.fs_segment_forward2 = wrapped  # instance attribute of the proxy: found before __getattr__
    stream = torch.cuda.current_stream()
    ev = torch.cuda.Event()

    def wait_stream():
        stream.synchronize()

    def wait_event():
        ev.record(stream)
        ev.synchronize()

    def wait_poll():
        ev.record(stream)
        while not ev.query():
            pass

    for name, wait in (("stream.synchronize", wait_stream), ("event.synchronize", wait_event), ("event.query spin", wait_poll)):
        rec = []
        for i in range(120):
            t0 = time.perf_counter()
            r = fm.predict(keys[i % 4:i % 4 + 1], keys[i % 4 + 1:i % 4 + 2], dl, dr, N, None, with_mask=True)
            t1 = time.perf_counter()
            host.copy_(r["mask"], non_blocking=True)
            t2 = time.perf_counter()
            wait()
            t3 = time.perf_counter()
            rec.append((marks["enter"] - t0, marks["leave"] - marks["enter"], t1 - marks["leave"], t2 - t1, t3 - t2, t3 - t0))
        rec = rec[20:]
        med = [statistics.median(c) * 1e6 for c in zip(*rec)]
        print(f"{name:20s} to-library {med[0]:6.1f} us | library call {med[1]:6.1f} | tail ops {med[2]:6.1f} | copy enqueue {med[3]:6.1f} | wait {med[4]:7.1f} | "
              f"step {med[5]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
