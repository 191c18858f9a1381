#!/usr/bin/env python3
"""What this box's HBM takes for write-only and copy streams of the decoder-batch size (torch's own fill / copy kernels): the ceiling
the fused feature tail's 0.66 GB of stores is measured against."""
import torch


def t(fn, it=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for mb in (33, 133, 265, 663, 1326):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda")
    y = torch.empty(n, device="cuda")
    f = t(lambda: x.fill_(1.0))
    c = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB  fill {f:7.1f} us = {mb / f:5.2f} TB/s written   copy {c:7.1f} us = {2 * mb / c:5.2f} TB/s moved ({mb / c:5.2f} written)")
