#!/usr/bin/env python3
"""Bring-up diagnostic for the GPU box: every kernel against the matching torch CPU op, all
cases run (no stop at first failure), a table goes to stdout and gpurun_out/diag.txt.
Usage: python tools/gpu_diag.py [--perf] [--skip-parity]
"""
import argparse
import os
import sys
import time
import traceback

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib, ops  # noqa: E402
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr  # noqa: E402

ROWS = []
DEV = "cuda"


def report(name, got, ref, tol):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    if got.shape != ref.shape:
        ROWS.append((name, "SHAPE", f"{tuple(got.shape)} vs {tuple(ref.shape)}"))
        return
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    ok = err <= tol * scale and bool(torch.isfinite(got).all())
    ROWS.append((name, "ok" if ok else "FAIL", f"max_abs_err={err:.3e} ref_max={scale:.3e} rel={err / scale:.2e}"))


def guarded(fn):
    def run(*a, **k):
        try:
            fn(*a, **k)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            ROWS.append((fn.__name__ + str(a), "EXC", repr(e)[:200]))
            traceback.print_exc()
    return run


@guarded
def conv_case(b, h, w, cin, cout, k, stride, pad, dil, relu, res, tile, slice_out=False):
    g = torch.Generator().manual_seed(h * 1000 + cin + cout + k + tile)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, None, stride, pad, dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    r = None
    if res:
        r = torch.randn(ref.shape, generator=g)
        ref = ref + r
    if relu:
        ref = ref.relu()
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    rd = r.to(DEV).contiguous(memory_format=torch.channels_last) if res else None
    if slice_out:
        lib = _lib.load()
        ho, wo = ref.shape[2], ref.shape[3]
        big = torch.full((b, ho, wo, cout + 64), -7.0, device=DEV)
        wp = torch.empty((cout, k, k, cin), device=DEV)
        check(lib.fs_pack_conv_weight(ptr(wt.to(DEV)), ptr(wp), cout, cin, k, k, stream_ptr()))
        view = big[..., 32:]
        scd, shd = sc.to(DEV), sh.to(DEV)
        check(lib.fs_conv2d_nhwc(ptr(xd), cin, ptr(wp), ptr(scd), ptr(shd), ptr(rd), cout, ptr(view), cout + 64, b, h, w, cin,
                                 cout, k, k, stride, pad, dil, int(relu), tile, stream_ptr()))
        torch.cuda.synchronize()
        got = big[..., 32:32 + cout].permute(0, 3, 1, 2)
        untouched = bool((big[..., :32] == -7.0).all() and (big[..., 32 + cout:] == -7.0).all())
        ROWS.append((f"conv slice untouched t{tile}", "ok" if untouched else "FAIL", ""))
    else:
        got = ops.conv2d_nhwc(xd, wt.to(DEV), sc.to(DEV), sh.to(DEV), rd, stride, pad, dil, relu, tile)
    report(f"conv b{b} {h}x{w} {cin}->{cout} k{k} s{stride} p{pad} d{dil} relu{int(relu)} res{int(res)} tile{tile}", got, ref, 2e-5)


@guarded
def stem_case(b, h, w, cout, k, stride, pad):
    lib = _lib.load()
    g = torch.Generator().manual_seed(k)
    x = torch.randn(b, 3, h, w, generator=g)
    wt = torch.randn(cout, 3, k, k, generator=g) * 0.2
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g) * 0.1
    ref = (F.conv2d(x, wt, None, stride, pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).relu()
    ho, wo = ref.shape[2], ref.shape[3]
    out = torch.empty((b, ho, wo, cout), device=DEV)
    whwio = wt.permute(2, 3, 1, 0).contiguous().to(DEV)
    xd, scd, shd = x.to(DEV), sc.to(DEV), sh.to(DEV)
    check(lib.fs_stem_conv_nchw(ptr(xd), ptr(whwio), ptr(scd), ptr(shd), ptr(out), b, h, w, cout, k, k, stride, pad, stream_ptr()))
    report(f"stem b{b} {h}x{w} 3->{cout} k{k} s{stride}", out.permute(0, 3, 1, 2), ref, 2e-5)


@guarded
def pool_cases():
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 128, 37, 41, generator=g)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    ref = F.max_pool2d(x, 3, 2, 1)
    out = torch.empty((2, ref.shape[2], ref.shape[3], 128), device=DEV)
    check(lib.fs_maxpool3x3s2_nhwc(ptr(xd), ptr(out), 2, 37, 41, 128, stream_ptr()))
    report("maxpool 37x41 c128", out.permute(0, 3, 1, 2), ref, 0)
    for (h, w) in ((23, 29), (90, 90)):
        x = torch.randn(2, 128, h, w, generator=g)
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
        for bin_ in (1, 2, 3, 6):
            ref = F.adaptive_avg_pool2d(x, bin_)
            out = torch.empty((2, bin_ * bin_, 128), device=DEV)
            check(lib.fs_adaptive_avgpool_nhwc(ptr(xd), 128, ptr(out), 2, h, w, 128, bin_, stream_ptr()))
            report(f"avgpool {h}x{w} bin{bin_}", out.view(2, bin_, bin_, 128).permute(0, 3, 1, 2), ref, 1e-5)


@guarded
def layout_cases():
    lib = _lib.load()
    x = torch.randn(2, 37, 50 * 3 + 1)
    xd = x.to(DEV)
    out = torch.empty((2, x.shape[2], 37), device=DEV)
    check(lib.fs_nchw_to_nhwc(ptr(xd), ptr(out), 2, 37, x.shape[2], stream_ptr()))
    report("nchw_to_nhwc", out, x.permute(0, 2, 1), 0)
    back = torch.empty_like(xd)
    check(lib.fs_nhwc_to_nchw(ptr(out), ptr(back), 2, 37, x.shape[2], stream_ptr()))
    report("nhwc_to_nchw", back, x, 0)


@guarded
def flow_cases():
    g = torch.Generator().manual_seed(9)
    for ac in (False, True):
        x = torch.randn(2, 5, 17, 23, generator=g)
        grid = torch.rand(2, 9, 11, 2, generator=g) * 2.6 - 1.3  # includes out-of-range -> border clamp
        ref = F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=ac)
        report(f"grid_sample nchw ac{int(ac)}", ops.grid_sample(x.to(DEV), grid.to(DEV), ac), ref, 1e-5)
        x = torch.randn(1, 128, 19, 21, generator=g)
        grid = torch.rand(1, 7, 9, 2, generator=g) * 2.4 - 1.2
        ref = F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=ac)
        got = ops.grid_sample(x.to(DEV).contiguous(memory_format=torch.channels_last), grid.to(DEV), ac)
        report(f"grid_sample nhwc ac{int(ac)}", got, ref, 1e-5)
        x = torch.randn(2, 5, 13, 17, generator=g)
        ref = F.interpolate(x, size=(41, 37), mode="bilinear", align_corners=ac)
        report(f"resize nchw up ac{int(ac)}", ops.resize_bilinear(x.to(DEV), (41, 37), ac), ref, 1e-5)
        ref = F.interpolate(x, size=(7, 9), mode="bilinear", align_corners=ac)
        report(f"resize nchw down ac{int(ac)}", ops.resize_bilinear(x.to(DEV), (7, 9), ac), ref, 1e-5)
        x = torch.randn(1, 64, 11, 12, generator=g)
        ref = F.interpolate(x, size=(23, 25), mode="bilinear", align_corners=ac)
        got = ops.resize_bilinear(x.to(DEV).contiguous(memory_format=torch.channels_last), (23, 25), ac)
        report(f"resize nhwc ac{int(ac)}", got, ref, 1e-5)
    a = torch.randn(3, 5, 31, 33, generator=g)
    b = torch.randn(3, 5, 31, 33, generator=g)
    report("blend", ops.blend(a.to(DEV), 0.6, b.to(DEV), 0.4), 0.6 * a + 0.4 * b, 1e-6)
    report("blend single", ops.blend(a.to(DEV), 0.6), 0.6 * a, 1e-6)


def ref_seg_tail(o, o_next, mvl, mvr, n, hw, no_warp):
    up = lambda t: F.interpolate(t, size=hw, mode="bilinear", align_corners=True) if t.shape[2:] != tuple(hw) else t  # noqa: E731
    warp = lambda t, m: t if no_warp else F.grid_sample(t, m.float(), mode="bilinear", padding_mode="border", align_corners=False)  # noqa: E731
    o = up(o)
    maps = [o]
    if o_next is not None:
        o_next = up(o_next)
        fwd, bwd = [], []
        cur = o
        for m in mvl:
            cur = warp(cur, m)
            fwd.append(up(cur))
        cur = o_next
        for m in mvr:
            cur = warp(cur, m)
            bwd.append(up(cur))
        for p in range(1, n):
            maps.append((n - p) / n * fwd[p - 1] + p / n * bwd[n - p - 1])
    return torch.cat(maps, 0)


@guarded
def seg_tail_cases():
    g = torch.Generator().manual_seed(11)
    for (n, h, hw, hg) in ((5, 12, (89, 97), 6), (3, 9, (65, 65), 4)):
        o = torch.randn(1, 5, h, h + 1, generator=g)
        o2 = torch.randn(1, 5, h, h + 1, generator=g)
        ident = torch.stack(torch.meshgrid(torch.linspace(-1, 1, hg + 1), torch.linspace(-1, 1, hg), indexing="xy"), -1)[None]
        mk = lambda: (ident + (torch.rand(ident.shape, generator=g) - 0.5) * 0.3)  # noqa: E731
        mvl = [mk() for _ in range(n - 1)]
        mvr = [mk() for _ in range(n - 1)]
        for no_warp in (True, False):
            ref = ref_seg_tail(o, o2, mvl, mvr, n, hw, no_warp)
            lg, mask = ops.seg_tail(o.to(DEV), o2.to(DEV), [m.to(DEV) for m in mvl], [m.to(DEV) for m in mvr], n, hw, no_warp,
                                    True, True)
            report(f"seg_tail n{n} no_warp{int(no_warp)} logits", lg, ref, 2e-6)
            agree = (mask.cpu() == ref.max(1)[1].to(torch.uint8)).float().mean().item()
            ROWS.append((f"seg_tail n{n} no_warp{int(no_warp)} mask", "ok" if agree > 0.9995 else "FAIL", f"agree={agree:.6f}"))
        ref1 = ref_seg_tail(o, None, [], [], n, hw, True)
        lg, _ = ops.seg_tail(o.to(DEV), None, [], [], n, hw, True, True, False)
        report(f"seg_tail n{n} single", lg, ref1, 2e-6)
    x = torch.randn(3, 5, 40, 44, generator=g)
    ref = F.interpolate(x, size=(67, 120), mode="bilinear", align_corners=True).max(1)[1].to(torch.uint8)
    got = ops.resize_argmax_u8(x.to(DEV), (67, 120)).cpu()
    ROWS.append(("resize_argmax", "ok" if (got == ref).float().mean().item() > 0.9995 else "FAIL",
                 f"agree={(got == ref).float().mean().item():.6f}"))
    got = ops.argmax_u8(x.to(DEV)).cpu()
    ROWS.append(("argmax", "ok" if bool((got == x.max(1)[1].to(torch.uint8)).all()) else "FAIL", ""))
    p = torch.randint(0, 5, (3, 50, 60), generator=g, dtype=torch.uint8)
    t = torch.randint(0, 5, (3, 50, 60), generator=g, dtype=torch.uint8)
    t[0, :5] = 255
    hist = ops.iou_hist(p.to(DEV), t.to(DEV), 5).cpu()
    pm = p.clone()
    pm[t == 255] = 255
    inter = torch.stack([((pm == k) & (t == k)).sum() for k in range(5)])
    ao = torch.stack([(pm == k).sum() for k in range(5)])
    at = torch.stack([(t == k).sum() for k in range(5)])
    okh = bool((hist[0] == inter).all() and (hist[1] == ao).all() and (hist[2] == at).all())
    ROWS.append(("iou_hist", "ok" if okh else "FAIL", str(hist.tolist())))


def time_conv(b, h, w, cin, cout, k, stride, pad, dil, tile, iters=5):
    lib = _lib.load()
    x = torch.randn(b, h, w, cin, device=DEV)
    wp = torch.randn(cout, k, k, cin, device=DEV) * 0.01
    ho = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    out = torch.empty(b, ho, wo, cout, device=DEV)

    def run():
        check(lib.fs_conv2d_nhwc(ptr(x), cin, ptr(wp), None, None, None, 0, ptr(out), cout, b, h, w, cin, cout, k, k, stride, pad,
                                 dil, 1, tile, stream_ptr()))
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
    return ms, fl / ms / 1e9


def perf_cases():
    shapes = [
        ("dec 3x3 4096->512 @90", (2, 90, 90, 4096, 512, 3, 1, 1, 1)),
        ("l4 3x3 d4 512->512 @90", (2, 90, 90, 512, 512, 3, 1, 4, 4)),
        ("l4 1x1 512->2048 @90", (2, 90, 90, 512, 2048, 1, 1, 0, 1)),
        ("l4 1x1 2048->512 @90", (2, 90, 90, 2048, 512, 1, 1, 0, 1)),
        ("l3 3x3 d2 256->256 @90", (2, 90, 90, 256, 256, 3, 1, 2, 2)),
        ("l3 1x1 256->1024 @90", (2, 90, 90, 256, 1024, 1, 1, 0, 1)),
        ("l3 1x1 1024->256 @90", (2, 90, 90, 1024, 256, 1, 1, 0, 1)),
        ("l2 3x3 128->128 @90", (2, 90, 90, 128, 128, 3, 1, 1, 1)),
        ("l1 3x3 64->64 @179", (2, 179, 179, 64, 64, 3, 1, 1, 1)),
        ("l1 1x1 64->256 @179", (2, 179, 179, 64, 256, 1, 1, 0, 1)),
        ("stem 3x3 64->64 @357", (2, 357, 357, 64, 64, 3, 1, 1, 1)),
        ("stem 3x3 64->128 @357", (2, 357, 357, 64, 128, 3, 1, 1, 1)),
    ]
    for name, sh in shapes:
        for tile in (1, 2, 3, 4):
            if sh[4] < 128 and tile in (1, 4):
                continue
            try:
                ms, tf = time_conv(*sh, tile)
                ROWS.append((f"perf {name} tile{tile}", "t", f"{ms:.3f} ms  {tf:.1f} TFLOP/s"))
            except Exception as e:  # noqa: BLE001
                ROWS.append((f"perf {name} tile{tile}", "EXC", repr(e)[:160]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--perf", action="store_true")
    ap.add_argument("--skip-parity", action="store_true")
    args = ap.parse_args()
    t0 = time.time()
    print("device:", torch.cuda.get_device_name(0), "| lib version", _lib.load().fs_version(), flush=True)
    if not args.skip_parity:
        for tile in (1, 2, 3, 4):
            conv_case(1, 19, 19, 64, 64, 1, 1, 0, 1, False, False, tile)
            conv_case(2, 37, 35, 64, 128, 3, 1, 1, 1, True, False, tile)
            conv_case(1, 45, 45, 128, 128, 3, 2, 1, 1, True, False, tile)
            conv_case(1, 23, 23, 256, 256, 3, 1, 2, 2, True, True, tile)
            conv_case(1, 23, 23, 512, 192, 3, 1, 4, 4, False, True, tile)
            conv_case(2, 45, 45, 256, 512, 1, 2, 0, 1, False, False, tile)
            conv_case(1, 30, 30, 96, 160, 3, 1, 1, 1, True, True, tile, slice_out=True)
        print(f"conv parity done {time.time() - t0:.1f}s", flush=True)
        stem_case(2, 65, 71, 64, 3, 2, 1)
        stem_case(1, 65, 71, 64, 7, 2, 3)
        pool_cases()
        layout_cases()
        flow_cases()
        seg_tail_cases()
        print(f"op parity done {time.time() - t0:.1f}s", flush=True)
    if args.perf:
        perf_cases()
    os.makedirs("gpurun_out", exist_ok=True)
    lines = [f"{s:5s} {n:70s} {d}" for (n, s, d) in ROWS]
    txt = "\n".join(lines)
    print(txt)
    with open("gpurun_out/diag.txt", "w") as f:
        f.write(txt + "\n")
    bad = [r for r in ROWS if r[1] in ("FAIL", "EXC", "SHAPE")]
    print(f"\n{len(ROWS)} rows, {len(bad)} bad, {time.time() - t0:.1f}s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
