#!/usr/bin/env python3
"""gemm_planes_bf16x3 (both GEMM operands as pre-split bf16 planes) on the network's GEMM shapes of a 713x713 window (B = 2), next to
the in-register split kernel (fs_conv2d_nhwc_split) where the shape is a plain 1x1 conv.  Random operands (the kernels are
power-limited: zeros would flatter them).  usage: plane_gemm_bench.py [iters] [dev]   (dev: the `make DEV=1` library, variants)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib  # noqa: E402

DEV = len(sys.argv) > 2 and sys.argv[2] == "dev"
if DEV:
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libfloodseg.so", "libfloodseg_dev.so")
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

SHAPES = {  # M, N, K, groups
    "head wino F(6,3) 2048->512": (450, 512, 2048, 64),
    "layer4.conv2 wino 512->512": (450, 512, 512, 64),
    "layer3.conv2 wino 256->256": (450, 256, 256, 64),
    "layer4.conv1 2048->512": (16200, 512, 2048, 1),
    "layer4.conv3 512->2048": (16200, 2048, 512, 1),
    "layer3.conv1 1024->256": (16200, 256, 1024, 1),
    "layer3.conv3 256->1024": (16200, 1024, 256, 1),
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


def planes(lib, t):
    out = torch.empty(3 * t.numel(), dtype=torch.bfloat16, device=t.device)
    check(lib.fs_split_bf16x3(ptr(t), t.numel(), ptr(out), stream_ptr()))
    return out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lib = _lib.load()
    variants = [0, 2 << 8] if DEV else [0]  # dev: 2 = the v_mfma_f32_16x16x32_bf16 variant
    print(f"{'shape':30s} {'in-register us':>14s} {'TF/s eq':>8s} | " + " | ".join(f"planes v{v >> 8} us  TF/s eq" for v in variants))
    for name, (M, N, K, G) in SHAPES.items():
        g = torch.Generator(device="cuda").manual_seed(1)
        a = torch.randn(G, M, K, device="cuda", generator=g).relu()
        w = torch.randn(G, N, K, device="cuda", generator=g) * (2.0 / K) ** 0.5
        a3, w3 = planes(lib, a), planes(lib, w)
        out = torch.empty(G, M, N, device="cuda")
        flops = 2.0 * G * M * N * K
        fns = {}
        if G == 1:
            o2 = torch.empty(M, N, device="cuda")

            def reg():
                check(lib.fs_conv2d_nhwc_split(ptr(a), K, ptr(w3), None, None, None, 0, ptr(o2), N, 1, M, 1, K, N, 1, 1, 1, 0, 1, 1, 0, stream_ptr()))
            fns["reg"] = reg
        for v in variants:
            def pl(v=v):
                check(lib.fs_gemm_bf16x3_planes(ptr(a3), a.numel(), K, ptr(w3), w.numel(), K, None, None, ptr(out), N, M, N, K, 1, G, M * K, N * K, M * N,
                                                v, stream_ptr()))
            fns[v] = pl
        # the kernels are power-limited and the clock the chip holds depends on what ran just before: the candidates take turns
        # (5 rounds of `iters` launches each, medians), never one after the other in a single long block
        times = {k: [] for k in fns}
        for _ in range(5):
            for k, fn in fns.items():
                times[k].append(timed(fn, iters))
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        t_reg = med.get("reg")
        cols = []
        for v in variants:
            t = med[v]
            cols.append(f"{t * 1e3:12.1f} {flops / t / 1e9:8.1f}")
            fns[v]()
            if G == 1:
                fns["reg"]()
                if v == 0:
                    assert torch.equal(out[0], o2), "plane kernel and in-register split kernel disagree"
                else:
                    err = ((out[0].double() - o2.double()).abs().max() / o2.abs().max()).item()
                    assert err < 5e-6, f"variant {v >> 8} differs from the in-register kernel by {err:.2e}"
                    cols[-1] += f" (vs v0 {err:.1e})"
        reg_s = f"{t_reg * 1e3:14.1f} {flops / t_reg / 1e9:8.1f}" if t_reg else f"{'-':>14s} {'-':>8s}"
        print(f"{name:30s} {reg_s} | " + " | ".join(cols))


if __name__ == "__main__":
    main()
