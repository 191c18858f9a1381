#!/usr/bin/env python3
"""Does a launch pay for following a DIFFERENT kernel (cold instruction cache, resource re-configuration)?  Two dependent-free GEMMs A
and B, timed as A A A ..., B B B ... and A B A B ..., once with two different tile instantiations and once with the same one."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402


def gemm(lib, m, cin, cout, tile, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(1, 1, m, cin, device="cuda", generator=g).relu()
    wp = torch.randn(cout, 1, 1, cin, device="cuda", generator=g) * (2.0 / cin) ** 0.5
    pl = torch.empty(3 * wp.numel(), dtype=torch.bfloat16, device="cuda")
    check(lib.fs_split_bf16x3(ptr(wp), wp.numel(), ptr(pl), stream_ptr()))
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    out = torch.empty(1, 1, m, cout, device="cuda")
    keep = (x, pl, sc, sh, out)

    def fn():
        check(lib.fs_conv2d_nhwc_split(ptr(x), cin, ptr(pl), ptr(sc), ptr(sh), None, cout, ptr(out), cout, 1, 1, m, cin, cout, 1, 1, 1, 0, 1, 0, tile,
                                       stream_ptr()))
    fn.keep = keep
    return fn


def timed(fns, n=240):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _lib.load()
    cases = [
        ("layer3 conv1 (128x64) / conv3 (128x128)", (16200, 1024, 256, 2), (16200, 256, 1024, 1)),
        ("layer3 conv1 / conv3, both 128x128", (16200, 1024, 256, 1), (16200, 256, 1024, 1)),
        ("vit qkv (128x96) / proj (64x64)", (4052, 384, 1152, 6), (4052, 384, 384, 3)),
        ("vit qkv / proj, both 64x64", (4052, 384, 1152, 3), (4052, 384, 384, 3)),
        ("vit qkv (128x96) / fc1 (128x96)", (4052, 384, 1152, 6), (4052, 384, 1536, 6)),
    ]
    print(f"{'pair':44s} {'A A A':>8s} {'B B B':>8s} {'mean':>8s} {'A B A B':>8s} {'switch cost':>12s}   us per launch, medians of 7")
    for name, a, b in cases:
        fa, fb = gemm(lib, *a, 1), gemm(lib, *b, 2)
        ta, tb, tab = [], [], []
        for _ in range(7):
            ta.append(timed([fa]))
            tb.append(timed([fb]))
            tab.append(timed([fa, fb]))
        ma, mb, mab = (statistics.median(v) for v in (ta, tb, tab))
        print(f"{name:44s} {ma:8.1f} {mb:8.1f} {(ma + mb) / 2:8.1f} {mab:8.1f} {mab - (ma + mb) / 2:+12.1f}", flush=True)


if __name__ == "__main__":
    main()
