#!/usr/bin/env python3
"""The fused Winograd F(4x4,3x3) kernel (wino_fused.hip) against the direct implicit-GEMM kernel on the network's small-Cin 3x3
layer shapes of a 713x713 window (B = 2).  usage: wino_fused_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # b, h, w, cin, cout
    "layer0.3": (2, 357, 357, 64, 64),
    "layer0.6": (2, 357, 357, 64, 128),
    "layer1.conv2": (2, 179, 179, 64, 64),
    "layer2.conv2": (2, 90, 90, 128, 128),
    "crops16 layer0.3": (16, 357, 357, 64, 64),
    "crops16 layer2.conv2": (16, 90, 90, 128, 128),
}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lib = _lib.load()
    print(f"{'shape':24s} {'direct us':>10s} {'TF/s':>7s} | {'fused v1 us':>11s} {'fused v2 us':>11s} {'fused v3 us':>11s} {'best eff. TF/s':>14s} {'MFMA TF/s':>10s}")
    for name, (b, h, w, cin, cout) in SHAPES.items():
        x = torch.randn(b, h, w, cin, device="cuda")
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wp = torch.empty(cout, 3, 3, cin, device="cuda")
        check(lib.fs_pack_conv_weight(ptr(wt), ptr(wp), cout, cin, 3, 3, stream_ptr()))
        out = torch.empty(b, h, w, cout, device="cuda")
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        ws = torch.empty(lib.fs_winograd_fused_workspace_floats(cin, cout), device="cuda")
        flops = 2.0 * b * h * w * cout * cin * 9

        def direct():
            check(lib.fs_conv2d_nhwc(ptr(x), cin, ptr(wp), ptr(sc), ptr(sh), None, 0, ptr(out), cout, b, h, w, cin, cout, 3, 3, 1, 1, 1, 1, 0, stream_ptr()))
        td = timed(direct, iters)
        tf = []
        for v in (1, 2, 3):
            def fused(v=v):
                check(lib.fs_conv3x3_winograd_fused_nhwc(ptr(x), cin, ptr(wt), ptr(sc), ptr(sh), ptr(out), cout, b, h, w, cin, cout, 1, v, ptr(ws), stream_ptr()))
            tf.append(timed(fused, iters))  # includes the (small) filter transform launch of the test entry
        best = min(tf)
        tiles = b * ((h + 3) // 4) * ((w + 3) // 4)
        mfma = 2.0 * 36 * tiles * cin * cout
        print(f"{name:24s} {td * 1e3:10.1f} {flops / td / 1e9:7.1f} | {tf[0] * 1e3:11.1f} {tf[1] * 1e3:11.1f} {tf[2] * 1e3:11.1f} {flops / best / 1e9:14.1f} {mfma / best / 1e9:10.1f}")


if __name__ == "__main__":
    main()
