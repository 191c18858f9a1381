#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one window (no synchronisation) compared with the GPU time of that window?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import ops, synth  # noqa: E402
from flood_uav_video_segmentation_amd.flow.model import FlowModel  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402

torch.set_grad_enabled(False)


class HP:
    layers, classes, pretrained = 50, 5, False


net = FlowPSPNet(HP()).eval()
net.load_state_dict(synth.make_pspnet_state(50, 5, seed=0))
fm = FlowModel(net, feature_based=False, no_warp=True).eval()
keys = synth.make_clip(6, 713, seed=1000, only=[0, 5]).cuda()
dl, dr = [[g.cuda() for g in gs] for gs in synth.dummy_grids(5)]
for _ in range(5):
    ops.argmax_u8(fm.predict(keys[0:1], keys[1:2], dl, dr, 5, None)["pred"])
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(n):
    ops.argmax_u8(fm.predict(keys[0:1], keys[1:2], dl, dr, 5, None)["pred"])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/window, GPU-bound total {1e3 * (t2 - t0) / n:.3f} ms/window")
