#!/usr/bin/env python3
"""Probe: the two key frames of a window as ONE batch of two on one stream (what FlowModel does) against two batch-1 forwards on two
streams with two library handles (same weights).  PSPNet-R50 @713.  usage: two_stream_probe.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import synth  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402


class HP:
    layers, classes, pretrained = 50, 5, False


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    torch.set_grad_enabled(False)
    state = synth.make_pspnet_state(50, 5, 0)
    a, b = FlowPSPNet(HP()).eval(), FlowPSPNet(HP()).eval()
    a.load_state_dict(state)
    b.load_state_dict(state)
    clip = synth.make_clip(6, 713, seed=1000, only=[0, 5]).cuda()
    prev, nxt = clip[0:1], clip[1:2]
    side = torch.cuda.Stream()

    def one_batch():
        return a.segment(prev, nxt)

    def two_streams():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        x = a.segment(prev)
        with torch.cuda.stream(side):
            y = b.segment(nxt)
        main_s.wait_stream(side)
        return x, y

    for name, fn in (("one batch of two, one stream", one_batch), ("two batch-1 forwards, two streams", two_streams)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        print(f"{name:36s} {(time.perf_counter() - t0) / iters * 1e3:7.3f} ms per key-frame pair")
    x, y = two_streams()
    both = one_batch()
    torch.cuda.synchronize()
    print("bit-identical:", torch.equal(both[0:1], x) and torch.equal(both[1:2], y))


if __name__ == "__main__":
    main()
