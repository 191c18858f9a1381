#!/bin/bash
# tile 6 (128 x 96) for the Segmenter's Linears: op tests, isolated sweep, per-launch table and configs[3] FPS against another build
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
OTHER=${1:-tools/bin/libfloodseg_r5b.so}
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "split_operands or chain" > $O/vt_pytest.txt 2>&1 || { tail -20 $O/vt_pytest.txt; exit 1; }
tail -2 $O/vt_pytest.txt
timeout -k 10 300 python tools/b1_tile_sweep.py 2 > $O/vt_sweep_b2.txt 2>&1 || { tail $O/vt_sweep_b2.txt; exit 1; }
grep -E "shape|vit" $O/vt_sweep_b2.txt
timeout -k 10 200 python tools/vit_profile.py s16 $OTHER > $O/vt_layers_other.txt 2>&1 || { tail $O/vt_layers_other.txt; exit 1; }
timeout -k 10 200 python tools/vit_profile.py s16 > $O/vt_layers_tree.txt 2>&1 || { tail $O/vt_layers_tree.txt; exit 1; }
echo "== other"; head -8 $O/vt_layers_other.txt; grep total $O/vt_layers_other.txt
echo "== tree"; head -8 $O/vt_layers_tree.txt; grep total $O/vt_layers_tree.txt
for i in 1 2 3; do
  timeout -k 10 200 python tools/bench_configs.py --only cfg3 --steps 40 --json --lib $OTHER 2>&1 | tail -1 | sed "s/^/other $i /"
  timeout -k 10 200 python tools/bench_configs.py --only cfg3 --steps 40 --json 2>&1 | tail -1 | sed "s/^/tree  $i /"
done
