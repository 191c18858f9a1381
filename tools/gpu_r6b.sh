#!/bin/bash
# round 6: the tests behind the pruned library + the fused qkv epilogue A/B on configs[3]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_vit.py -q -m gpu -x --durations=8 > $O/pytest.txt 2>&1; echo "pytest exit=$?"; tail -14 $O/pytest.txt
for i in 1 2 3; do
python tools/bench_configs.py --only cfg3 --steps 60 2>&1 | grep -v amdgpu.ids | tail -1
python tools/bench_configs.py --only cfg3 --steps 60 --opt hip_no_fused_qkv 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/$/  [no fused qkv]/'
done | tee $O/ab.txt
python tools/vit_profile.py s16 2>&1 | grep -v amdgpu.ids > $O/vit_s16_layers.txt; head -14 $O/vit_s16_layers.txt
