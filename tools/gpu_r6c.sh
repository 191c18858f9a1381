#!/bin/bash
# round 6: bench.py's new host-side fields + the 4-core squeeze test; B = 1 against B = 2 layer tables; DeepLab F(4,3) / F(6,3) forced
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -q -m gpu -x --durations=8 > $O/pytest_bench.txt 2>&1; echo "pytest exit=$?"; tail -14 $O/pytest_bench.txt
python tools/layer_profile.py 1 2>&1 | grep -v amdgpu.ids > $O/layers_b1.txt; tail -16 $O/layers_b1.txt
python tools/layer_profile.py 2 2>&1 | grep -v amdgpu.ids > $O/layers_b2.txt; tail -16 $O/layers_b2.txt
for o in "" "--opt hip_winograd_tile=4" "--opt hip_winograd_tile=6"; do
python tools/bench_configs.py --only cfg2 --steps 30 $o 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/\$/  [$o]/"
done | tee $O/cfg2_wino_tile.txt
timeout -k 10 300 python tools/b1_tile_sweep.py 1 2>&1 | grep -v amdgpu.ids > $O/b1_tile_sweep.txt; tail -5 $O/b1_tile_sweep.txt
