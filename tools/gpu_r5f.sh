#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "winograd_fused or fused_stem" > $O/r5f_pytest.txt 2>&1; rc=$?
tail -4 $O/r5f_pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/wino_fused_bench.py 100 > $O/r5f_wino_fused_bench.txt 2>&1; tail -12 $O/r5f_wino_fused_bench.txt
for opt in "" hip_wino_dma; do
  timeout -k 10 300 python tools/layer_profile.py 2 pspnet50 $opt > $O/r5f_layers_${opt:-default}.txt 2>&1 || exit 1
  echo "== ${opt:-default}"; grep -E "layer0|total" $O/r5f_layers_${opt:-default}.txt
done
tools/gpu_ab_opts.sh 150 - hip_wino_dma
