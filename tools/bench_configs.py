#!/usr/bin/env python3
"""FPS of every BASELINE.json config on ONE MI355X (secondary numbers for DESIGN.md; bench.py is the headline).
All fp32, synthetic weights/frames, inputs resident in HBM, uint8 masks copied to the host each step."""
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
    def __init__(self, layers):
        self.layers, self.classes, self.pretrained = layers, 5, False


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
    dev = "cuda"
    host = torch.empty((N, 713, 713), dtype=torch.uint8).pin_memory()
    rows = []
    keys = synth.make_clip(21, 713, seed=1000, only=[0, 5, 10, 15, 20]).to(dev)
    dl, dr = [[g.to(dev) for g in gs] for gs in synth.dummy_grids(N)]
    wl, wr = [[g.to(dev) for g in gs] for gs in synth.make_grids(N, 44, 44, seed=2000)]

    def window(fm, grids):
        def step(i):
            out = fm.predict(keys[i % 4:i % 4 + 1], keys[i % 4 + 1:i % 4 + 2], grids[0], grids[1], N, None)["pred"]
            host.copy_(ops.argmax_u8(out), non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return step

    psp = FlowPSPNet(HP(50)).eval()
    psp.load_state_dict(synth.make_pspnet_state(50, 5, 0))

    def single(i):  # configs[0] semantics on the GPU: one frame per step, PSPNet.forward + argmax
        lo = psp.segment(keys[i % 5:i % 5 + 1])
        _, mask = ops.seg_tail(lo, None, [], [], 1, (713, 713), True, want_logits=False, want_mask=True)
        host[:1].copy_(mask, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    t = timeit(single)
    rows.append(("configs[0] PSPNet-R50 single-frame (GPU)", 1 / t, t * 1e3))
    t = timeit(window(FlowModel(psp, feature_based=False, no_warp=True).eval(), (dl, dr)))
    rows.append(("configs[1] PSPNet-R50 keyframe + linear interp", N / t, t * 1e3))
    t = timeit(window(FlowModel(psp, feature_based=False, no_warp=False).eval(), (wl, wr)))
    rows.append(("configs[4]/1GPU PSPNet-R50 keyframe + logit warp", N / t, t * 1e3))
    t = timeit(window(FlowModel(psp, feature_based=True, no_warp=False).eval(), (wl, wr)), steps=5)
    rows.append(("(extra) PSPNet-R50 keyframe + FEATURE warp", N / t, t * 1e3))
    del psp
    dl3 = FlowDeepLabv3(HP(101)).eval()
    dl3.load_state_dict(synth.make_deeplab_state(101, 5, 0))
    t = timeit(window(FlowModel(dl3, feature_based=False, no_warp=False).eval(), (wl, wr)))
    rows.append(("configs[2] DeepLabv3-R101 keyframe + logit warp", N / t, t * 1e3))
    del dl3
    vit = VITSegmentModel(5, 704, patch_size=16, d_model=384, n_layers=12, dec_layers=2).eval()
    vit.load_state_dict(synth.make_vit_state(5, 704, 16, 384, 12, 2, seed=0))
    t = timeit(window(FlowModel(vit, feature_based=True, no_warp=False).eval(), (wl, wr)))
    rows.append(("configs[3] Segmenter ViT-S/16 keyframe + feature flow (extension)", N / t, t * 1e3))
    vitb = VITSegmentModel(5, 704).eval()
    vitb.load_state_dict(synth.make_vit_state(5, 704, seed=0))

    def vit_single(i):
        out = vitb(keys[i % 5:i % 5 + 1])["pred"]
        host[:1].copy_(ops.argmax_u8(out.contiguous()), non_blocking=True)
        torch.cuda.current_stream().synchronize()
    t = timeit(vit_single)
    rows.append(("(extra) Segmenter ViT-B/32 per-frame (as model/vit.py builds it)", 1 / t, t * 1e3))
    print(f"{'config':68s} {'FPS':>9s} {'ms/step':>9s}")
    for name, fps, ms in rows:
        print(f"{name:68s} {fps:9.1f} {ms:9.3f}")


if __name__ == "__main__":
    main()
