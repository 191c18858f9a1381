#!/bin/bash
# Round 5: the chained conv3 -> next conv1 launch (conv_chain_dma_f32): parity first, then the layer table and an interleaved A/B of
# the whole window against the round-4 library on the same box.  usage: tools/gpu_chain.sh [steps]
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "chain or option" > $O/r5_chain_pytest.txt 2>&1; rc=$?
tail -5 $O/r5_chain_pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/layer_profile.py 2 > $O/r5_layers_b2_chain.txt 2>&1 || exit 1
grep -E "chain|total|layer1|layer2" $O/r5_layers_b2_chain.txt | head -40
[ -f tools/bin/libfloodseg_r4.so ] && tools/gpu_ab_bench.sh tools/bin/libfloodseg_r4.so ${1:-150}
