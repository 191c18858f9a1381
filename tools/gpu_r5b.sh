#!/bin/bash
# Round 5, batch B: fused stem max-pool + residual touch -- parity, layer tables for each route, whole-window A/B against the round-4 library.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_net.py -m gpu -q -x -p no:cacheprovider -k "pool or option or touch or stem or reserve or graph" > $O/r5b_pytest.txt 2>&1; rc=$?
tail -5 $O/r5b_pytest.txt
[ $rc -ne 0 ] && exit $rc
for opt in "" hip_no_fused_pool hip_res_touch; do
  timeout -k 10 300 python tools/layer_profile.py 2 pspnet50 $opt > $O/r5b_layers_${opt:-default}.txt 2>&1 || exit 1
  echo "== ${opt:-default}"; grep -E "layer0|maxpool|total|^  " $O/r5b_layers_${opt:-default}.txt | head -24
done
[ -f tools/bin/libfloodseg_r4.so ] && tools/gpu_ab_bench.sh tools/bin/libfloodseg_r4.so ${1:-150}
