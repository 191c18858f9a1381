#!/bin/bash
# Kernel stats + PMC passes (SQ sets, FETCH_SIZE, WRITE_SIZE, TCC hits: separate passes, kernel-trace only) for the fused
# predict_feature tail alone (tools/feat_tail_bench.py).  Output: gpurun_out/pmc_feat/summary.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_feat
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/feat_tail_bench.py 10 > $O/stats.log 2>&1; echo "stats exit=$?"
f=$(find $O/stats -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -8 "$f" | cut -c1-150
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/p$i -- python3 $R/tools/feat_tail_bench.py 4 > $O/p$i.log 2>&1; echo "pass $i ($SET) exit=$?"
done
python3 - <<PY > $O/summary.txt
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "feat_" in k:
                agg[(k.split("(")[0].replace("void fs::", "").replace("fs::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"{k:28s} {c:36s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
PY
cat $O/summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
