#!/usr/bin/env python3
"""Print ONE window of a rocprofv3 --kernel-trace csv: every dispatch between two consecutive occurrences of a marker kernel
(default: the fused tail, which ends a window), with its duration and the idle gap before it.

    rocprofv3 --kernel-trace -d gpurun_out/trace -o w -- python3 bench.py --steps 8 --warmup 2
    python tools/trace_window.py gpurun_out/trace/w_results.db (or a *_kernel_trace.csv) [--marker seg_fuse_kernel] [--which -2]
"""
import argparse
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?((?:\w+::)*\w+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="seg_fuse_kernel")
    ap.add_argument("--which", type=int, default=-2, help="index of the marker occurrence that STARTS the window printed")
    a = ap.parse_args()
    if a.csv.endswith(".db"):  # rocprofv3's default output on this image: a rocpd sqlite database with a `kernels` view
        import sqlite3
        q = "select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"
        rows = [dict(zip(("Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X"), r))
                for r in sqlite3.connect(a.csv).execute(q)]
    else:
        rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit(f"marker {a.marker!r} seen {len(marks)} times")
    lo, hi = marks[a.which] + 1, marks[a.which + 1] + 1 if a.which != -1 else len(rows)
    prev_end = int(rows[lo - 1]["End_Timestamp"])
    busy = idle = 0
    for n, r in enumerate(rows[lo:hi]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = s - prev_end
        busy += e - s
        idle += max(gap, 0)
        print(f"{n:4d} {short(r['Kernel_Name']):70s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}x{int(r['Grid_Size_Y']):4d}x{int(r['Grid_Size_Z']):3d}"
              f"  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}")
        prev_end = max(prev_end, e)
    print(f"# {hi - lo} dispatches, busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us")


if __name__ == "__main__":
    main()
