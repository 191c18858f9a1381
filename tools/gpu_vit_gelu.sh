#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
OTHER=${1:-tools/bin/libfloodseg_r5b.so}
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_flow.py -m gpu -q -x -p no:cacheprovider -k "gelu or vit or segmenter or Segmenter or config3" > $O/vg_pytest.txt 2>&1 || { tail -30 $O/vg_pytest.txt; exit 1; }
tail -2 $O/vg_pytest.txt
timeout -k 10 200 python tools/vit_profile.py s16 > $O/vg_layers_tree.txt 2>&1 || { tail $O/vg_layers_tree.txt; exit 1; }
head -8 $O/vg_layers_tree.txt; grep total $O/vg_layers_tree.txt
for i in 1 2 3; do
  timeout -k 10 200 python tools/bench_configs.py --only cfg3 --steps 40 --json --lib $OTHER 2>&1 | tail -1 | sed "s/^/other $i /"
  timeout -k 10 200 python tools/bench_configs.py --only cfg3 --steps 40 --json 2>&1 | tail -1 | sed "s/^/tree  $i /"
done
