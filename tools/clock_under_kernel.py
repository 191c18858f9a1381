#!/usr/bin/env python3
"""Why do isolated kernel wins vanish inside a window?  One GEMM launched back to back for ~1.5 s per candidate tile while the card's shader clock
and socket power are sampled (bench.ClockSampler): time per launch, clock held, power drawn, and time x clock (cycles per launch).
usage: clock_under_kernel.py [seconds per candidate] [other build of the library]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # rows, cin, cout, tiles to compare
    "layer4 conv1 16200 x 2048 -> 512": (16200, 2048, 512, (1, 2)),
    "layer4 conv3 16200 x 512 -> 2048": (16200, 512, 2048, (1, 2)),
    "layer3 conv1 16200 x 1024 -> 256": (16200, 1024, 256, (2, 1, 3)),
}
NAMES = {1: "128x128", 2: "128x64", 3: "64x64"}  # (the 256 x 128 tiles of profiles/r05_experiments.txt section 19 were removed in round 6)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    if len(sys.argv) > 2:
        _lib.LIB_PATH, _lib.ALLOW_MISSING = os.path.abspath(sys.argv[2]), True
    lib = _lib.load()
    dev = len(sys.argv) > 2
    print(f"{'shape':36s} {'tile':>18s} {'us':>8s} {'MHz':>7s} {'W':>7s} {'k cycles':>9s} {'J / launch':>10s}")
    for name, (m, cin, cout, tiles) in SHAPES.items():
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(1, 1, m, cin, device="cuda", generator=g).relu()
        wp = torch.randn(cout, 1, 1, cin, device="cuda", generator=g) * (2.0 / cin) ** 0.5
        pl = torch.empty(3 * wp.numel(), dtype=torch.bfloat16, device="cuda")
        check(lib.fs_split_bf16x3(ptr(wp), wp.numel(), ptr(pl), stream_ptr()))
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        out = torch.empty(1, 1, m, cout, device="cuda")
        for rep in range(2):
            for tile in tiles:

                def fn():
                    check(lib.fs_conv2d_nhwc_split(ptr(x), cin, ptr(pl), ptr(sc), ptr(sh), None, cout, ptr(out), cout, 1, 1, m, cin, cout, 1, 1, 1, 0, 1, 0, tile,
                                                   stream_ptr()))
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
                n = 0
                with bench.ClockSampler(torch) as cs:
                    t0 = time.perf_counter()
                    while time.perf_counter() - t0 < secs:
                        for _ in range(200):
                            fn()
                        torch.cuda.synchronize()
                        n += 200
                    dt = time.perf_counter() - t0
                us = dt / n * 1e6
                mhz = sum(cs.clk) / max(len(cs.clk), 1)
                w = sum(cs.pw) / max(len(cs.pw), 1)
                print(f"{name:36s} {NAMES[tile]:>18s} {us:8.1f} {mhz:7.0f} {w:7.0f} {us * mhz / 1e3:9.1f} {w * us * 1e-6:10.4f}", flush=True)


if __name__ == "__main__":
    main()
