#!/usr/bin/env python3
"""fs_feat_tail alone (the fused predict_feature tail) on the PSPNet feature geometry of 713^2 frames: C = 4096, 90 x 90 maps,
44 x 44 grids, n = 5 -- per mode, HIP-event time per call and the bytes it has to move.

    python tools/feat_tail_bench.py [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import ops, synth  # noqa: E402
from flood_uav_video_segmentation_amd.flow.model import get_default_grid  # noqa: E402

torch.set_grad_enabled(False)


def timed(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    if len(sys.argv) > 2:  # development A/B: another build of the library
        from flood_uav_video_segmentation_amd import _lib
        _lib.LIB_PATH, _lib.ALLOW_MISSING = os.path.abspath(sys.argv[2]), True
    for C, fh in ((4096, 90),):
        n, hg = 5, 44
        f = torch.randn(1, C, fh, fh, device="cuda").contiguous(memory_format=torch.channels_last)
        g = torch.randn(1, C, fh, fh, device="cuda").contiguous(memory_format=torch.channels_last)
        mvl, mvr = [[m.cuda() for m in ms] for ms in synth.make_grids(n, hg, hg, seed=2000)]
        g0 = torch.from_numpy(get_default_grid()).float().unsqueeze(0).cuda()
        mb = C * fh * fh * 4 / 1e6
        rows = [("warp n=5 (chains + 5 maps)", lambda: ops.feat_tail(f, g, mvl, mvr, n, False, g0), 5 * mb + 8 * 2 * C * hg * hg * 4 / 1e6 + 2 * mb),
                ("warp, single frame (map 0 only)", lambda: ops.feat_tail(f, None, mvl, mvr, n, False, g0), 2 * mb),
                ("no_warp n=5 (5 maps)", lambda: ops.feat_tail(f, g, mvl, mvr, n, True), 7 * mb),
                ("no_warp single (copy)", lambda: ops.feat_tail(f, None, mvl, mvr, n, True), 2 * mb)]
        print(f"C={C} map {fh}x{fh} ({mb:.1f} MB), grids {hg}x{hg}")
        for name, fn, mbytes in rows:
            us = timed(fn, iters)
            print(f"  {name:36s} {us:8.1f} us   {mbytes:8.1f} MB min -> {mbytes / us:6.2f} TB/s")


if __name__ == "__main__":
    main()
