#!/bin/bash
# PMC passes for one conv shape (separate passes: SQ set, FETCH_SIZE, WRITE_SIZE), kernel-trace only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc
SHAPE=${1:-dec}
TILE=${2:-0}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/conv_bench.py $SHAPE $TILE 10 > $O/plain.txt 2>&1; cat $O/plain.txt | tail -2
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/p$i -- python3 $R/tools/conv_bench.py $SHAPE $TILE 5 > $O/p$i.log 2>&1; echo "pass $i ($SET) exit=$?"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "conv_igemm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
PY
find $O -name "*kernel_trace.csv" -delete
