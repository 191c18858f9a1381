#!/bin/bash
# Idle time between the kernels of one window: rocprofv3 --kernel-trace of a short bench run, then per timed step the sum of the
# gaps between consecutive kernel executions on the device.  -> gpurun_out/gaps.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/gaps; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; echo "exit=$?"
python3 - <<PY > $R/gpurun_out/gaps.txt
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps = runs of kernels starting with the stem conv
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "stem_conv" in n]
out = []
for a, b in zip(starts[4:14], starts[5:15]):
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]), seg[i]["Kernel_Name"][:40], seg[i + 1]["Kernel_Name"][:40]) for i in range(len(seg) - 1)]
    out.append((t1 - t0, busy, len(seg), gaps, int(rows[b]["Start_Timestamp"]) - t1))
for span, busy, n, gaps, to_next in out:
    print(f"window: {n} kernels, first start -> last end {span/1e3:.1f} us, sum of kernel durations {busy/1e3:.1f} us, idle inside {100*(span-busy)/span:.1f} %, last end -> next window's first start {to_next/1e3:.1f} us")
span, busy, n, gaps, _ = out[3]
print("gaps of one window (us) after -> before:")
import collections
agg = collections.defaultdict(list)
for g, a, b in gaps:
    agg[(a, b)].append(g / 1e3)
for (a, b), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"  {sum(v):7.1f} total  x{len(v):<3d} avg {sum(v)/len(v):5.2f}   {a} -> {b}")
print("total gap", sum(g for g, _, _ in gaps) / 1e3, "us over", len(gaps), "boundaries")
PY
rm -rf $O
cat $R/gpurun_out/gaps.txt | head -45
