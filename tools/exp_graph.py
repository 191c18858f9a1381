#!/usr/bin/env python3
"""Experiment (GPU box): one window's launch sequence replayed from a captured HIP graph versus enqueued eagerly -- is there
any inter-kernel gap a graph would close?  (Host enqueue takes 0.3 ms per 6 ms window, so none is expected.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import ops, synth  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402

torch.set_grad_enabled(False)


class HP:
    layers, classes, pretrained = 50, 5, False


def main():
    net = FlowPSPNet(HP()).eval()
    net.load_state_dict(synth.make_pspnet_state(50, 5, 0))
    keys = synth.make_clip(6, 713, seed=1000, only=[0, 5]).cuda()
    dl, dr = [[g.cuda() for g in gs] for gs in synth.dummy_grids(5)]

    def window():
        lows = net.segment(keys[0:1], keys[1:2])
        _, mask = ops.seg_tail(lows[0:1], lows[1:2], dl, dr, 5, (713, 713), True, want_logits=False, want_mask=True)
        return mask

    for _ in range(5):
        ref = window()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        window()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) * 10
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        window()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = window()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) * 10
    print(f"eager {eager:.3f} ms/window   graph replay {graph:.3f} ms/window   (network + fused tail, no host copy)")


if __name__ == "__main__":
    main()
