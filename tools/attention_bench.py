#!/usr/bin/env python3
"""fs_attention on the Segmenter's shapes: the pipelined split-operand kernel (1) against the stage-serial one (2) and the fp32 route (0);
bit-identity of 1 and 2.  usage: attention_bench.py [other build of libfloodseg.so ...]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402


def main():
    if len(sys.argv) > 1:
        _lib.LIB_PATH, _lib.ALLOW_MISSING = os.path.abspath(sys.argv[1]), True
    lib = _lib.load()
    print(f"library: {_lib.LIB_PATH if len(sys.argv) > 1 else 'in tree'}")
    for b, n, heads in ((2, 2026, 6), (1, 2026, 6), (4, 2026, 6), (2, 530, 12), (5, 2030, 6)):
        g = torch.Generator(device="cuda").manual_seed(n)
        qkv = torch.randn(b, n, 3 * heads * 64, device="cuda", generator=g) * 1.5
        outs, med = {}, {}
        for mode in (1, 2, 0):
            ws = torch.empty(lib.fs_attention_workspace_floats(b, n, heads, mode), device="cuda")
            out = torch.empty(b, n, heads * 64, device="cuda")

            def fn():
                check(lib.fs_attention(ptr(qkv), ptr(out), b, n, heads, 0.125, mode, ptr(ws), stream_ptr()))
            for _ in range(5):
                fn()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 50 * 1e3)
            med[mode] = statistics.median(ts)
            outs[mode] = out.clone()
        gf = 4.0 * b * heads * n * n * 64 * 1e-9
        print(f"B={b} N={n} heads={heads}: stage-serial {med[1]:7.1f} us ({gf / med[1] * 1e3:6.1f} TFLOP/s-eq)   pipelined {med[2]:7.1f} us ({gf / med[2] * 1e3:6.1f})   fp32 {med[0]:7.1f} us   "
              f"max |1 - 2| {float((outs[1] - outs[2]).abs().max()):.2e}   max |1 - fp32| {float((outs[1] - outs[0]).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
