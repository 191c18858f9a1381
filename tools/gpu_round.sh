#!/bin/bash
# One complete GPU-box round: parity tests (measured errors -> gpurun_out/parity_measured.txt), smoke, the default bench
# (-> gpurun_out/bench_n1.json), rocprofv3 kernel stats for configs 1/2/3 + layer tables (tools/gpu_profiles.sh) and the
# PMC traffic passes (tools/gpu_pmc_bench.sh).  usage: GIT_HEAD=<short sha> tools/gpu_round.sh [noprofile]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/parity_measured.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit=$rc"; tail -3 $O/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke exit=$?"; tail -1 $O/smoke.txt
timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench exit=$?"; cut -c1-400 $O/bench_n1.json
[ "$1" == "noprofile" ] && exit 0
tools/gpu_profiles.sh && tools/gpu_pmc_bench.sh


