#!/usr/bin/env python3
"""FPS of every BASELINE.json config on ONE MI355X (secondary numbers for DESIGN.md; bench.py is the headline).
All fp32, synthetic weights/frames, inputs resident in HBM, uint8 masks copied to the host each step.

    python tools/bench_configs.py                 # all rows
    python tools/bench_configs.py --only cfg2     # one config (cfg0 cfg1 cfg4 feat cfg2 cfg3 vitb) -- the command that
                                                  # `rocprofv3 --kernel-trace --stats` wraps for profiles/r02_cfg*_kernel_stats.csv
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import ops, synth  # noqa: E402
from flood_uav_video_segmentation_amd.flow.model import FlowModel  # noqa: E402
from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402
from flood_uav_video_segmentation_amd.model.vit import VITSegmentModel  # noqa: E402

torch.set_grad_enabled(False)
N = 5


class HP:
    OPTIONS = ()   # --opt words: hip_* attributes of the reference-style hparams (model/hipnet.py::hip_options)

    def __init__(self, layers):
        self.layers, self.classes, self.pretrained = layers, 5, False
        for word in HP.OPTIONS:  # NAME or NAME=INT
            name, _, val = word.partition("=")
            setattr(self, name, int(val) if val else True)


def timeit(fn, steps=10, warmup=2):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="cfg0 | cfg1 | cfg4 | feat | crops | crops_cached | cfg2 | cfg3 | vitb")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--opt", action="append", default=[], help="hip_no_split_bf16 | hip_no_winograd | hip_winograd_tile=4 | ... (repeatable; model/hipnet.py::HIP_OPTIONS)")
    ap.add_argument("--lib", default=None, help="development A/B: load this build of the library instead of the in-tree one")
    ap.add_argument("--vit-two-step", action="store_true", help="A/B: the Segmenter's upsample, unpadding and argmax as separate steps (cfg3)")
    ap.add_argument("--feat-op-by-op", action="store_true", help="A/B: predict_feature's tail op by op instead of fs_feat_tail (feat, cfg3)")
    ap.add_argument("--json", action="store_true", help="also print one JSON line {value, unit, ms_per_step, steps} of the last config run")
    args = ap.parse_args()
    HP.OPTIONS = tuple(args.opt)
    if args.lib:
        from flood_uav_video_segmentation_amd import _lib
        _lib.LIB_PATH, _lib.ALLOW_MISSING = os.path.abspath(args.lib), True
    want = lambda k: not args.only or args.only == k  # noqa: E731
    dev = "cuda"
    host = torch.empty((N, 713, 713), dtype=torch.uint8).pin_memory()
    rows = []
    keys = synth.make_clip(21, 713, seed=1000, only=[0, 5, 10, 15, 20]).to(dev)
    dl, dr = [[g.to(dev) for g in gs] for gs in synth.dummy_grids(N)]
    wl, wr = [[g.to(dev) for g in gs] for gs in synth.make_grids(N, 44, 44, seed=2000)]

    def window(fm, grids):
        fm.fused_feature_tail = not args.feat_op_by_op

        def step(i):
            r = fm.predict(keys[i % 4:i % 4 + 1], keys[i % 4 + 1:i % 4 + 2], grids[0], grids[1], N, None, with_mask=True)
            host.copy_(r["mask"], non_blocking=True)  # logits AND masks are produced; the masks go to the host
            torch.cuda.current_stream().synchronize()
        return step

    psp = None
    if any(want(k) for k in ("cfg0", "cfg1", "cfg4", "feat", "crops", "crops_cached")):
        psp = FlowPSPNet(HP(50)).eval()
        psp.load_state_dict(synth.make_pspnet_state(50, 5, 0))

    def single(i):  # configs[0] semantics on the GPU: one frame per step, PSPNet.forward + argmax
        lo = psp.segment(keys[i % 5:i % 5 + 1])
        _, mask = ops.seg_tail(lo, None, [], [], 1, (713, 713), True, want_logits=False, want_mask=True)
        host[:1].copy_(mask, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    st = args.steps
    if want("cfg0"):
        t = timeit(single, st)
        rows.append(("configs[0] PSPNet-R50 single-frame (GPU)", 1 / t, t * 1e3))
    if want("cfg1"):
        t = timeit(window(FlowModel(psp, feature_based=False, no_warp=True).eval(), (dl, dr)), st)
        rows.append(("configs[1] PSPNet-R50 keyframe + linear interp", N / t, t * 1e3))
    if want("cfg4"):
        t = timeit(window(FlowModel(psp, feature_based=False, no_warp=False).eval(), (wl, wr)), st)
        rows.append(("configs[4]/1GPU PSPNet-R50 keyframe + logit warp", N / t, t * 1e3))
    if want("feat"):
        t = timeit(window(FlowModel(psp, feature_based=True, no_warp=False).eval(), (wl, wr)), steps=max(1, st // 2))
        rows.append(("(extra) PSPNet-R50 keyframe + FEATURE warp", N / t, t * 1e3))
    for cached in (False, True):
        if not want("crops_cached" if cached else "crops"):
            continue
        # the reference's default real-video route (flow/base.py:182-209): 1072x1920 frames, 8 overlapping 713x713 crops of both key
        # frames, warp, float64 canvas, masks at 1072x1920 (bench.py's fps_real_video_route_* variants, on their own for rocprofv3)
        from flood_uav_video_segmentation_amd.flow.predict import FlowPredictor
        hd = synth.make_clip(16, (1072, 1920), seed=1200, only=[0, 5, 10, 15]).to(dev)
        gl, gr = [[g.to(dev) for g in gs] for gs in synth.make_grids(N, 67, 120, seed=2100, frame=(1072, 1920), jitter=0.01)]
        pred = FlowPredictor(FlowModel(psp, feature_based=False, no_warp=False).eval(), 5, (1072, 1920), crop=(713, 713), compute_metrics=False,
                             cache_keyframes=cached)
        psp.reserve(16, 713, 713)
        host_hd = torch.empty((N, 1072, 1920), dtype=torch.uint8).pin_memory()

        def crops_step(i, pred=pred, cached=cached):
            w = i % 3
            if cached and w == 0:
                pred.reset()
            host_hd.copy_(pred.predict_window(hd[w:w + 1], hd[w + 1:w + 2], gl, gr, to_host=False, key_ids=(5 * w, 5 * w + 5) if cached else None),
                          non_blocking=True)
            torch.cuda.current_stream().synchronize()
        t = timeit(crops_step, steps=max(3, st))
        rows.append(("(reference default route) 1072x1920, 8 crops x 2 key frames, warp" + (", key-frame cache" if cached else ""), N / t, t * 1e3))
        del pred, hd
    del psp
    if want("cfg2"):
        dl3 = FlowDeepLabv3(HP(101)).eval()
        dl3.load_state_dict(synth.make_deeplab_state(101, 5, 0))
        t = timeit(window(FlowModel(dl3, feature_based=False, no_warp=False).eval(), (wl, wr)), st)
        rows.append(("configs[2] DeepLabv3-R101 keyframe + logit warp", N / t, t * 1e3))
        del dl3
    if want("cfg3"):
        vit = VITSegmentModel(5, 704, patch_size=16, d_model=384, n_layers=12, dec_layers=2, **{w.partition("=")[0]: (int(w.partition("=")[2]) if "=" in w else True) for w in HP.OPTIONS}).eval()
        vit.load_state_dict(synth.make_vit_state(5, 704, 16, 384, 12, 2, seed=0))
        if args.vit_two_step:
            vit.decode_fit = None  # FlowModel._decode_fit then takes decoder -> fit_output -> argmax_u8 one by one
        t = timeit(window(FlowModel(vit, feature_based=True, no_warp=False).eval(), (wl, wr)), st)
        rows.append(("configs[3] Segmenter ViT-S/16 keyframe + feature flow (extension)", N / t, t * 1e3))
        del vit
    if want("vitb"):
        vitb = VITSegmentModel(5, 704, **{w.partition("=")[0]: (int(w.partition("=")[2]) if "=" in w else True) for w in HP.OPTIONS}).eval()
        vitb.load_state_dict(synth.make_vit_state(5, 704, seed=0))

        def vit_single(i):
            out = vitb(keys[i % 5:i % 5 + 1])["pred"]
            host[:1].copy_(ops.argmax_u8(out), non_blocking=True)
            torch.cuda.current_stream().synchronize()
        t = timeit(vit_single, st)
        rows.append(("(extra) Segmenter ViT-B/32 per-frame (as model/vit.py builds it)", 1 / t, t * 1e3))
    print(f"{'config':68s} {'FPS':>9s} {'ms/step':>9s}")
    for name, fps, ms in rows:
        print(f"{name:68s} {fps:9.1f} {ms:9.3f}")
    if args.json and rows:
        import json
        print(json.dumps({"config": rows[-1][0], "options": list(HP.OPTIONS), "value": round(rows[-1][1], 2), "unit": "frames/s",
                          "ms_per_step": round(rows[-1][2], 4), "steps": st}))


if __name__ == "__main__":
    main()
