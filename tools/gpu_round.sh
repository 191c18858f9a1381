#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$1" != "noprofile" ]; then
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.txt 2>&1; echo "pytest exit=$?"; tail -3 $O/pytest_gpu.txt
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke exit=$?"; tail -2 $O/smoke.txt
fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"; cat $O/bench.json; tail -3 $O/bench.err
if [ "$1" != "noprofile" ]; then
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof.err; echo "rocprof exit=$?"
find $O/prof -name "*kernel_stats*.csv" | head -3
f=$(find $O/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep the merged output small: the per-dispatch trace is large
find $O/prof -name "*kernel_trace*.csv" -size +20M -delete
fi
