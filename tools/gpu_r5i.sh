#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
tools/gpu_chunks.sh > $O/r5i_chunks.txt 2>&1; grep -E "==|chunk [0-3]|prologue|TFLOP" $O/r5i_chunks.txt
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "conv or repeat or 713 or chain or option or vit" > $O/r5i_pytest.txt 2>&1; rc=$?
tail -4 $O/r5i_pytest.txt
[ $rc -ne 0 ] && exit $rc
tools/gpu_ab_bench.sh tools/bin/libfloodseg_r5b.so ${1:-150}
