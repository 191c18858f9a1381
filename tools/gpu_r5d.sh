#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
rm -f $O/parity_measured.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/r5d_pytest.txt 2>&1; rc=$?
tail -4 $O/r5d_pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/b1_tile_sweep.py 2 > $O/r5d_tile_sweep_b2.txt 2>&1 || exit 1
cat $O/r5d_tile_sweep_b2.txt
timeout -k 10 300 python tools/vit_profile.py s16 > $O/r5d_vit_s16_layers.txt 2>&1; tail -16 $O/r5d_vit_s16_layers.txt
[ -f tools/bin/libfloodseg_r4.so ] && tools/gpu_ab_bench.sh tools/bin/libfloodseg_r4.so ${1:-150}
