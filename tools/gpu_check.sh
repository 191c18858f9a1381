#!/bin/bash
# Quick GPU round for kernel work: the tests that touch the conv / stem / shortcut / crops paths, then the per-layer tables.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 \
  -k "stem or net or shortcut or option or crops or two_tensor or segment_crops or conv" > gpurun_out/r02_pytest_quick.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_pytest_quick.txt; [ $rc -ne 0 ] && exit $rc
python tools/layer_profile.py 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_layers_b2_cur.txt
grep -E "layer0|conv3\+downsample|downsample|total|  igemm|  stem|  wino" gpurun_out/r02_layers_b2_cur.txt
python tools/layer_profile.py 2 deeplab101 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_layers_deeplab101_cur.txt
grep -E "backbone.conv1|aspp.pool|total|  igemm|  stem|  wino|  adaptive" gpurun_out/r02_layers_deeplab101_cur.txt
