#!/usr/bin/env python3
"""Summarise a tools/probe_conv_trace CSV: where a workgroup's cycles go and how workgroups share CUs."""
import sys
from collections import defaultdict

import numpy as np


def main(path):
    head = open(path).readline().strip()
    d = np.genfromtxt(path, delimiter=",", skip_header=2, dtype=np.uint64)
    if d.ndim == 1:
        d = d[None]
    bid, start, ready, loop, end, wait, hwid, xcc, w0, w1 = [d[:, i].astype(np.int64) for i in range(10)]
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    key = xcc * 4096 + se * 64 + sh * 32 + cu
    print(head)
    n = len(bid)
    print(f"workgroups {n}, distinct CUs {len(set(key.tolist()))}, XCCs {sorted(set(xcc.tolist()))}")
    tot = end - start
    f = lambda a: f"mean {a.mean():9.0f}  p10 {np.percentile(a, 10):9.0f}  p50 {np.percentile(a, 50):9.0f}  p90 {np.percentile(a, 90):9.0f}  max {a.max():9.0f}"  # noqa: E731
    print("cycles per workgroup (shader clock):")
    print("  total        ", f(tot))
    print("  prologue+DMA0", f(ready - start))
    print("  main loop    ", f(loop - ready))
    print("  epilogue     ", f(end - loop))
    print("  wait+barrier ", f(wait), f"= {100.0 * wait.sum() / (loop - ready).sum():.1f} % of the main loop")
    wall = (w1.max() - w0.min()) * 10.0  # ns (100 MHz)
    print(f"wall span first start -> last end: {wall / 1000:.1f} us")
    # shader clock rate estimate from workgroups: cycles / wall ns
    ok = (w1 - w0) > 100
    if ok.any():
        ghz = (tot[ok] / ((w1 - w0)[ok] * 10.0)).mean()
        print(f"shader clock ~ {ghz:.3f} GHz (s_memtime cycles per wall ns)")
    per = defaultdict(list)
    for i in range(n):
        per[int(key[i])].append(i)
    counts = np.array([len(v) for v in per.values()])
    print(f"workgroups per CU: min {counts.min()} max {counts.max()} mean {counts.mean():.2f}; histogram {np.bincount(counts).tolist()}")
    # per-CU busy span and co-residency
    spans, overl = [], []
    for v in per.values():
        s, e = w0[v], w1[v]
        spans.append((e.max() - s.min()) * 10.0)
        ev = sorted([(int(a), 1) for a in s] + [(int(b), -1) for b in e])
        cur, last, acc = 0, ev[0][0], defaultdict(int)
        for tt, dd in ev:
            acc[cur] += tt - last
            cur, last = cur + dd, tt
        busy = sum(acc[k] for k in acc if k > 0)
        overl.append(sum(k * acc[k] for k in acc) / max(busy, 1))
    spans = np.array(spans)
    print(f"per-CU busy span (us): mean {spans.mean() / 1000:.1f}  min {spans.min() / 1000:.1f}  max {spans.max() / 1000:.1f};  mean co-resident workgroups while busy {np.mean(overl):.2f}")
    starts = (w0 - w0.min()) * 10.0
    print(f"start skew (us): p50 {np.percentile(starts, 50) / 1000:.1f}  p90 {np.percentile(starts, 90) / 1000:.1f}  max {starts.max() / 1000:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
