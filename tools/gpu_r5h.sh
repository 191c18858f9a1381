#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
rm -f $O/parity_measured.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/r5h_pytest.txt 2>&1; rc=$?
tail -4 $O/r5h_pytest.txt
[ $rc -ne 0 ] && exit $rc
tools/gpu_ab_bench.sh tools/bin/libfloodseg_r4.so ${1:-150}
python tools/bench_configs.py > $O/r5h_all_configs.txt 2>&1; grep -v amdgpu $O/r5h_all_configs.txt
