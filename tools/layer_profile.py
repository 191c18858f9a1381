#!/usr/bin/env python3
"""Per-launch table (HIP events inside the library) of one key-frame batch at 713x713:
    python tools/layer_profile.py [B=2] [pspnet50 | deeplab101] [hip_no_winograd ...]   (further words: hparams options set True; model/hipnet.py::HIP_OPTIONS)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import synth  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402


class HP:
    def __init__(self, layers):
        self.layers, self.classes, self.pretrained = layers, 5, False
        for opt in sys.argv[3:]:
            name, _, val = opt.partition("=")
            setattr(self, name, int(val) if val else True)


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    arch = sys.argv[2] if len(sys.argv) > 2 else "pspnet50"
    torch.set_grad_enabled(False)
    if arch == "deeplab101":
        from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3
        net = FlowDeepLabv3(HP(101)).eval()
        net.load_state_dict(synth.make_deeplab_state(101, 5, 0))
    else:
        net = FlowPSPNet(HP(50)).eval()
        net.load_state_dict(synth.make_pspnet_state(50, 5, 0))
    x = synth.make_clip(b, 713, seed=1000).cuda()
    for _ in range(2):
        net.segment(x)
    torch.cuda.synchronize()
    reps = 5
    net._hip_net.profile(True)
    for _ in range(reps):
        net.segment(x)
    rows = net._hip_net.profile_dump()
    nops = len(rows) // reps
    tot = 0.0
    print(f"{'op':42s} {'kernel':18s} {'ms':>8s} {'GFLOP':>9s} {'TFLOP/s':>8s}")
    for i in range(nops):
        ms = sum(rows[i + r * nops][4] for r in range(reps)) / reps
        name, kernel, flops = rows[i][0], rows[i][1], rows[i][2]
        tot += ms
        print(f"{name:42s} {kernel:18s} {ms:8.4f} {flops / 1e9:9.3f} {flops / ms / 1e9 if ms > 0 else 0:8.1f}")
    print(f"total {tot:.3f} ms for B={b} ({arch})")
    per = {}
    for i in range(nops):
        ms = sum(rows[i + r * nops][4] for r in range(reps)) / reps
        d = per.setdefault(rows[i][1], [0.0, 0.0, 0])
        d[0] += ms
        d[1] += rows[i][2]
        d[2] += 1
    for k, (ms, fl, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:18s} x{n:<3d} {ms:8.4f} ms  {fl / 1e9:9.2f} GFLOP  {fl / ms / 1e9 if ms > 0 else 0:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
