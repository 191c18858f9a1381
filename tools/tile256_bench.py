#!/usr/bin/env python3
"""DEV experiment (make DEV=1 library): the split-operand kernel with a 256 x 128 workgroup tile -- 4 x 1 waves of 64 x 128 (tile 5, one
workgroup per CU, accumulators in AGPRs) and, round 5, 8 x 1 waves of 32 x 128 (tile 7, 512 threads) -- against the shipped 128 x 128 tile (two workgroups per CU) on the long-K layer shapes.
Candidates interleaved, medians of REPS blocks of ITERS launches.  usage: tile256_bench.py [iters] [reps]"""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402

_lib.LIB_PATH = _lib.LIB_PATH.replace("libfloodseg.so", "libfloodseg_dev.so")
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # b, h, w, cin, cout, residual
    "layer4.conv1 2048->512": (2, 90, 90, 2048, 512, False),
    "layer4.conv3 512->2048 +res": (2, 90, 90, 512, 2048, True),
    "layer3.conv3 256->1024 +res": (2, 90, 90, 256, 1024, True),
    "layer4.0.conv1 1024->512": (2, 90, 90, 1024, 512, False),
    "aspp 1x1 2048->256 (128x64 default)": (2, 90, 90, 2048, 256, False),
    "layer3.conv1 1024->256": (2, 90, 90, 1024, 256, False),
    "layer2.conv3 128->512 +res": (2, 90, 90, 128, 512, True),
}


def block(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = _lib.load()
    print(f"{'shape':38s} {'tile':>12s} {'us':>8s} {'TF/s eq':>8s} {'err/max':>9s}")
    for name, (b, h, w, cin, cout, res) in SHAPES.items():
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(b, h, w, cin, device="cuda", generator=g).relu()
        wt = torch.randn(cout, cin, 1, 1, device="cuda", generator=g) * (2.0 / cin) ** 0.5
        r = torch.randn(b, h, w, cout, device="cuda", generator=g) if res else None
        wp = torch.empty(cout, 1, 1, cin, device="cuda")
        check(lib.fs_pack_conv_weight(ptr(wt), ptr(wp), cout, cin, 1, 1, stream_ptr()))
        planes = torch.empty(3 * wp.numel(), dtype=torch.bfloat16, device="cuda")
        check(lib.fs_split_bf16x3(ptr(wp), wp.numel(), ptr(planes), stream_ptr()))
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double()).permute(0, 2, 3, 1)
        ref = ((ref + r.double()) if res else ref).relu()
        flops = 2.0 * b * h * w * cout * cin
        outs, fns = {}, {}
        for tile in (0, 1, 5, 7):
            outs[tile] = torch.empty(b, h, w, cout, device="cuda")

            def fn(tile=tile):
                check(lib.fs_conv2d_nhwc_split(ptr(x), cin, ptr(planes), ptr(sc), ptr(sh), ptr(r), cout, ptr(outs[tile]), cout, b, h, w, cin, cout,
                                               1, 1, 1, 0, 1, 1, tile, stream_ptr()))
            fns[tile] = fn
            fn()
        times = {t: [] for t in fns}
        for _ in range(reps):
            for t, fn in fns.items():
                times[t].append(block(fn, iters))
        for t in fns:
            ms = statistics.median(times[t])
            err = ((outs[t].double() - ref).abs().max() / ref.abs().max()).item()
            same = "" if t == 0 else ("  bit-identical to tile 0" if torch.equal(outs[t], outs[0]) else "  differs from tile 0")
            print(f"{name:38s} {({0: 'auto', 1: '128x128', 5: '256x128 4w', 7: '256x128 8w'})[t]:>12s} {ms * 1e3:8.1f} {flops / ms / 1e9:8.1f} {err:9.2e}{same}")


if __name__ == "__main__":
    main()
