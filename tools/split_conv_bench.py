#!/usr/bin/env python3
"""The split-operand conv kernel (conv_igemm_dma_f32<..., SPLIT>: fp32 = three bf16 terms, six cross products on the bf16 matrix
cores) against the fp32-MFMA kernel on the network's layer shapes of a 713x713 window (B = 2): time, and the error of BOTH against a
float64 convolution of the same fp32 inputs.  usage: split_conv_bench.py [iters]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # b, h, w, cin, cout, k, dil, residual
    "layer4.conv1 2048->512": (2, 90, 90, 2048, 512, 1, 1, False),
    "layer4.conv3 512->2048 +res": (2, 90, 90, 512, 2048, 1, 1, True),
    "layer3.conv1 1024->256": (2, 90, 90, 1024, 256, 1, 1, False),
    "layer3.conv3 256->1024 +res": (2, 90, 90, 256, 1024, 1, 1, True),
    "layer2.conv3 128->512 +res": (2, 90, 90, 128, 512, 1, 1, True),
    "layer1.conv1 256->64": (2, 179, 179, 256, 64, 1, 1, False),
    "layer3.conv2 3x3 d2 256": (2, 90, 90, 256, 256, 3, 2, False),
    "vit qkv 384->1152": (1, 1, 4052, 384, 1152, 1, 1, False),
    "vit fc1 384->1536": (1, 1, 4052, 384, 1536, 1, 1, False),
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
    print(f"{'shape':30s} {'tile':>4s} {'fp32 us':>9s} {'TF/s':>7s} {'err/max':>9s} | {'split us':>9s} {'TF/s eq':>8s} {'err/max':>9s} {'speedup':>7s}")
    for name, (b, h, w, cin, cout, k, dil, res) in SHAPES.items():
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(b, h, w, cin, device="cuda", generator=g).relu()  # post-ReLU activations
        wt = torch.randn(cout, cin, k, k, device="cuda", generator=g) * (2.0 / (cin * k * k)) ** 0.5
        r = torch.randn(b, h, w, cout, device="cuda", generator=g) if res else None
        wp = torch.empty(cout, k, k, cin, device="cuda")
        check(lib.fs_pack_conv_weight(ptr(wt), ptr(wp), cout, cin, k, k, stream_ptr()))
        planes = torch.empty(3 * wp.numel(), dtype=torch.bfloat16, device="cuda")
        check(lib.fs_split_bf16x3(ptr(wp), wp.numel(), ptr(planes), stream_ptr()))
        pl = planes.view(3, -1).float()
        assert torch.equal(pl[0] + pl[1] + pl[2], wp.view(-1)), "the three bf16 terms do not add up to the fp32 filters"
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        pad = dil * (k // 2)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, 1, pad, dil).permute(0, 2, 3, 1)
        if res:
            ref = ref + r.double()
        ref = ref.relu()
        flops = 2.0 * b * h * w * cout * cin * k * k
        for tile in (0, 1, 2, 3):
            o1, o2 = torch.empty(b, h, w, cout, device="cuda"), torch.empty(b, h, w, cout, device="cuda")

            def f32():
                check(lib.fs_conv2d_nhwc(ptr(x), cin, ptr(wp), ptr(sc), ptr(sh), ptr(r), cout, ptr(o1), cout, b, h, w, cin, cout, k, k, 1, pad, dil, 1,
                                         tile, stream_ptr()))

            def split():
                check(lib.fs_conv2d_nhwc_split(ptr(x), cin, ptr(planes), ptr(sc), ptr(sh), ptr(r), cout, ptr(o2), cout, b, h, w, cin, cout, k, k, 1, pad,
                                               dil, 1, tile, stream_ptr()))
            t1, t2 = timed(f32, iters), timed(split, iters)
            e1 = ((o1.double() - ref).abs().max() / ref.abs().max()).item()
            e2 = ((o2.double() - ref).abs().max() / ref.abs().max()).item()
            print(f"{name:30s} {tile:4d} {t1 * 1e3:9.1f} {flops / t1 / 1e9:7.1f} {e1:9.2e} | {t2 * 1e3:9.1f} {flops / t2 / 1e9:8.1f} {e2:9.2e} {t1 / t2:7.2f}")


if __name__ == "__main__":
    main()
