#!/usr/bin/env python3
"""Experiment (GPU box): one window's two key frames as ONE batch of two on one stream (the shipped route) versus one frame
per stream on two library handles (kernels of the two frames overlap: one's tile prologue / epilogue / launch tail under the
other's main loop).  Prints ms per window for both."""
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
    dev = "cuda"
    state = synth.make_pspnet_state(50, 5, 0)
    a, b = FlowPSPNet(HP()).eval(), FlowPSPNet(HP()).eval()
    a.load_state_dict(state)
    b.load_state_dict(state)
    keys = synth.make_clip(6, 713, seed=1000, only=[0, 5]).to(dev)
    dl, dr = [[g.to(dev) for g in gs] for gs in synth.dummy_grids(5)]
    side = torch.cuda.Stream()
    host = torch.empty((5, 713, 713), dtype=torch.uint8).pin_memory()

    def tail(lo_p, lo_n):
        _, mask = ops.seg_tail(lo_p, lo_n, dl, dr, 5, (713, 713), True, want_logits=False, want_mask=True)
        host.copy_(mask, non_blocking=True)
        torch.cuda.current_stream().synchronize()

    def batched(_):
        lows = a.segment(keys[0:1], keys[1:2])
        tail(lows[0:1], lows[1:2])

    def two_streams(_):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        lo_p = a.segment(keys[0:1])
        with torch.cuda.stream(side):
            lo_n = b.segment(keys[1:2])
        main.wait_stream(side)
        tail(lo_p, lo_n)

    def single(_):
        tail(a.segment(keys[0:1]), None)

    for name, fn in (("batched B=2, one stream", batched), ("B=1 + B=1 on two streams", two_streams), ("B=1 alone", single)):
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            fn(i)
        torch.cuda.synchronize()
        print(f"{name:32s} {(time.perf_counter() - t0) * 10:.3f} ms per window", flush=True)
    lows = a.segment(keys[0:1], keys[1:2])
    assert torch.equal(lows[0:1], a.segment(keys[0:1])) and torch.equal(lows[1:2], b.segment(keys[1:2]))
    print("bit-identical: yes")


if __name__ == "__main__":
    main()
