#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_fullsize.py tests/test_gpu_flow.py -m gpu -q -x -p no:cacheprovider -k "stem or 713 or option or crop or deeplab or float64" > $O/r5e_pytest.txt 2>&1; rc=$?
tail -4 $O/r5e_pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/layer_profile.py 2 > $O/r5e_layers.txt 2>&1 || exit 1
grep -E "layer0|total" $O/r5e_layers.txt
timeout -k 10 300 python tools/layer_profile.py 2 deeplab101 > $O/r5e_layers_dl.txt 2>&1 || exit 1
grep -E "conv1.weight  |stem|total" $O/r5e_layers_dl.txt | head -5
tools/gpu_ab_bench.sh tools/bin/libfloodseg_r5a.so ${1:-150}
