#!/usr/bin/env python3
"""Times one conv shape through fs_conv2d_nhwc (for rocprofv3 --pmc passes and tile A/B tests).
usage: conv_bench.py [shape-name] [tile] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # b, h, w, cin, cout, k, stride, pad, dil
    "dec": (2, 90, 90, 4096, 512, 3, 1, 1, 1),
    "l4c2": (2, 90, 90, 512, 512, 3, 1, 4, 4),
    "l4c3": (2, 90, 90, 512, 2048, 1, 1, 0, 1),
    "l4c1": (2, 90, 90, 2048, 512, 1, 1, 0, 1),
    "l3c2": (2, 90, 90, 256, 256, 3, 1, 2, 2),
    "l3c3": (2, 90, 90, 256, 1024, 1, 1, 0, 1),
    "l3c1": (2, 90, 90, 1024, 256, 1, 1, 0, 1),
    "l2c2": (2, 90, 90, 128, 128, 3, 1, 1, 1),
    "l1c3": (2, 179, 179, 64, 256, 1, 1, 0, 1),
    "stem3": (2, 357, 357, 64, 128, 3, 1, 1, 1),
    "aspp12": (2, 90, 90, 2048, 256, 3, 1, 12, 12),
    "aspp24": (2, 90, 90, 2048, 256, 3, 1, 24, 24),
    "aspp36": (2, 90, 90, 2048, 256, 3, 1, 36, 36),
}


def main():
    if os.environ.get("CONV_BENCH_B"):
        for k, v in list(SHAPES.items()):
            SHAPES[k] = (int(os.environ["CONV_BENCH_B"]),) + v[1:]
    name = sys.argv[1] if len(sys.argv) > 1 else "dec"
    tile = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lib = _lib.load()
    names = list(SHAPES) if name == "all" else [name]
    for nm in names:
        b, h, w, cin, cout, k, stride, pad, dil = SHAPES[nm]
        x = torch.randn(b, h, w, cin, device="cuda")
        wp = torch.randn(cout, k, k, cin, device="cuda") * 0.01
        ho = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
        wo = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
        out = torch.empty(b, ho, wo, cout, device="cuda")
        sc = torch.ones(cout, device="cuda")
        sh = torch.zeros(cout, device="cuda")

        def run():
            check(lib.fs_conv2d_nhwc(ptr(x), cin, ptr(wp), ptr(sc), ptr(sh), None, 0, ptr(out), cout, b, h, w, cin, cout, k, k, stride,
                                     pad, dil, 1, tile, stream_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * b * ho * wo * cout * k * k * cin
        alg_bytes = 4.0 * (b * h * w * cin + cout * k * k * cin + b * ho * wo * cout)
        print(f"{nm:6s} tile{tile & 255} korder{(tile >> 10) & 1} {ms:8.4f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  alg_bytes={alg_bytes / 1e6:.1f} MB", flush=True)


if __name__ == "__main__":
    main()
