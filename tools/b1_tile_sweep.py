#!/usr/bin/env python3
"""Tile sweep of the split-operand kernel on the layer shapes of ONE key frame (B = 1 @713: 8100-pixel maps) -- does the cost model
(pick_tile, tile 0) pick the fastest tile when a launch has few tiles?  Candidates interleaved, medians.  usage: b1_tile_sweep.py [B]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # h, w, cin, cout, k, stride, residual
    "layer1.0.conv1 128->64": (179, 179, 128, 64, 1, 1, False),
    "layer1.conv1 256->64": (179, 179, 256, 64, 1, 1, False),
    "layer1.conv3 64->256 +res": (179, 179, 64, 256, 1, 1, True),
    "layer2.0.conv1 256->128 @179": (179, 179, 256, 128, 1, 1, False),
    "layer2.0.conv2 3x3 s2 128->128": (179, 179, 128, 128, 3, 2, False),
    "layer2.conv1 512->128": (90, 90, 512, 128, 1, 1, False),
    "layer2.conv3 128->512 +res": (90, 90, 128, 512, 1, 1, True),
    "layer3.0.conv1 512->256": (90, 90, 512, 256, 1, 1, False),
    "layer3.conv1 1024->256": (90, 90, 1024, 256, 1, 1, False),
    "layer3.conv3 256->1024 +res": (90, 90, 256, 1024, 1, 1, True),
    "layer4.0.conv1 1024->512": (90, 90, 1024, 512, 1, 1, False),
    "layer4.conv1 2048->512": (90, 90, 2048, 512, 1, 1, False),
    "layer4.conv3 512->2048 +res": (90, 90, 512, 2048, 1, 1, True),
    "vit qkv 384->1152 (2026 tokens per frame)": (1, 2026, 384, 1152, 1, 1, False),
    "vit proj 384->384 +res": (1, 2026, 384, 384, 1, 1, True),
    "vit fc1 384->1536": (1, 2026, 384, 1536, 1, 1, False),
    "vit fc2 1536->384 +res": (1, 2026, 1536, 384, 1, 1, True),
}
NAMES = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x64", 4: "64x128", 6: "128x96"}


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
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    lib = _lib.load()
    print(f"B = {b}")
    print(f"{'shape':42s} " + " ".join(f"{NAMES[t]:>10s}" for t in NAMES) + "   best vs auto")
    only = sys.argv[2] if len(sys.argv) > 2 else ""   # substring filter on the shape names
    for name, (h, w, cin, cout, k, stride, res) in SHAPES.items():
        if only not in name:
            continue
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(b, h, w, cin, device="cuda", generator=g).relu()
        wt = torch.randn(cout, cin, k, k, device="cuda", generator=g) * (2.0 / (cin * k * k)) ** 0.5
        ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        r = torch.randn(b, ho, wo, cout, device="cuda", generator=g) if res else None
        wp = torch.empty(cout, k, k, cin, device="cuda")
        check(lib.fs_pack_conv_weight(ptr(wt), ptr(wp), cout, cin, k, k, stream_ptr()))
        planes = torch.empty(3 * wp.numel(), dtype=torch.bfloat16, device="cuda")
        check(lib.fs_split_bf16x3(ptr(wp), wp.numel(), ptr(planes), stream_ptr()))
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        out = torch.empty(b, ho, wo, cout, device="cuda")
        fns = {}
        for tile in NAMES:
            if (tile in (1, 4) and cout < 128) or (tile == 6 and cout % 96):
                continue

            def fn(tile=tile):
                check(lib.fs_conv2d_nhwc_split(ptr(x), cin, ptr(planes), ptr(sc), ptr(sh), ptr(r), cout, ptr(out), cout, b, h, w, cin, cout, k, k, stride, k // 2, 1, 1,
                                               tile, stream_ptr()))
            fns[tile] = fn
            fn()
        times = {t: [] for t in fns}
        for _ in range(5):
            for t, fn in fns.items():
                times[t].append(block(fn, 100))
        med = {t: statistics.median(v) * 1e3 for t, v in times.items()}
        best = min((t for t in med if t), key=lambda t: med[t])
        print(f"{name:42s} " + " ".join(f"{med[t]:10.1f}" if t in med else f"{'-':>10s}" for t in NAMES) + f"   {NAMES[best]} {100 * (med[0] / med[best] - 1):+.1f} %")


if __name__ == "__main__":
    main()
