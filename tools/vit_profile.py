#!/usr/bin/env python3
"""Per-launch table of one Segmenter forward (B=2). usage: vit_profile.py [s16|b32] [other build of libfloodseg.so, for an A/B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import _lib, synth  # noqa: E402
from flood_uav_video_segmentation_amd.model.vit import VITSegmentModel  # noqa: E402

torch.set_grad_enabled(False)
which = sys.argv[1] if len(sys.argv) > 1 else "s16"
if len(sys.argv) > 2 and sys.argv[2] != "-":
    _lib.LIB_PATH, _lib.ALLOW_MISSING = os.path.abspath(sys.argv[2]), True
cfg = dict(patch=16, d=384) if which == "s16" else dict(patch=32, d=768)
opts = {w: True for w in sys.argv[3:]}  # e.g. hip_no_fused_qkv
net = VITSegmentModel(5, 704, patch_size=cfg["patch"], d_model=cfg["d"], **opts).eval()
net.load_state_dict(synth.make_vit_state(5, 704, cfg["patch"], cfg["d"], 12, 2, seed=0))
x = synth.make_clip(2, 713, seed=3).cuda()
for _ in range(2):
    net(x)
torch.cuda.synchronize()
net._hip_net.profile(True)
reps = 3
for _ in range(reps):
    net(x)
rows = net._hip_net.profile_dump()
agg = {}
for name, kernel, flops, nbytes, ms in rows:
    key = (name.split(".")[-1] if "blocks" in name else name, kernel)
    d = agg.setdefault(key, [0.0, 0.0, 0])
    d[0] += ms / reps
    d[1] += flops / reps
    d[2] += 1
tot = sum(v[0] for v in agg.values())
for (name, kernel), (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:28s} {kernel:16s} {ms:8.4f} ms {fl / 1e9:9.2f} GF {fl / ms / 1e9 if ms else 0:7.1f} TF/s  x{n // reps}")
print(f"total {tot:.3f} ms (B=2, {which})")
