#!/usr/bin/env python3
"""Does a short-K launch pay for cold operands?  The same split-operand GEMM timed three ways, interleaved: one operand set reused (what
a tile sweep measures), NSETS filter banks in rotation (the weights of a stack of transformer blocks), and NSETS filter banks AND NSETS
pixel maps in rotation.  usage: cold_operands.py [nsets]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # rows, cin, cout
    "vit qkv 4052 x 384 -> 1152": (4052, 384, 1152),
    "vit fc1 4052 x 384 -> 1536": (4052, 384, 1536),
    "vit proj 4052 x 384 -> 384": (4052, 384, 384),
    "layer3 conv3 16200 x 256 -> 1024": (16200, 256, 1024),
    "layer3 conv1 16200 x 1024 -> 256": (16200, 1024, 256),
    "layer4 conv1 16200 x 2048 -> 512": (16200, 2048, 512),
}


def main():
    nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    lib = _lib.load()
    print(f"{'shape':36s} {'one set':>10s} {'filters x' + str(nsets):>12s} {'both x' + str(nsets):>12s}   us per launch (median of 5 x 240)")
    for name, (m, cin, cout) in SHAPES.items():
        g = torch.Generator(device="cuda").manual_seed(1)
        xs = [torch.randn(1, 1, m, cin, device="cuda", generator=g).relu() for _ in range(nsets)]
        planes = []
        for _ in range(nsets):
            wp = torch.randn(cout, 1, 1, cin, device="cuda", generator=g) * (2.0 / cin) ** 0.5
            pl = torch.empty(3 * wp.numel(), dtype=torch.bfloat16, device="cuda")
            check(lib.fs_split_bf16x3(ptr(wp), wp.numel(), ptr(pl), stream_ptr()))
            planes.append(pl)
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        outs = [torch.empty(1, 1, m, cout, device="cuda") for _ in range(2)]

        def launch(i, x, pl):
            check(lib.fs_conv2d_nhwc_split(ptr(x), cin, ptr(pl), ptr(sc), ptr(sh), None, cout, ptr(outs[i & 1]), cout, 1, 1, m, cin, cout, 1, 1, 1, 0, 1, 0, 0,
                                           stream_ptr()))
        modes = {"one": lambda i: launch(i, xs[0], planes[0]), "filters": lambda i: launch(i, xs[0], planes[i % nsets]),
                 "both": lambda i: launch(i, xs[i % nsets], planes[i % nsets])}
        times = {k: [] for k in modes}
        for _ in range(5):
            for k, fn in modes.items():
                for i in range(nsets):
                    fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(240):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 240 * 1e3)
        med = {k: statistics.median(v) for k, v in times.items()}
        print(f"{name:36s} {med['one']:10.1f} {med['filters']:12.1f} {med['both']:12.1f}", flush=True)


if __name__ == "__main__":
    main()
