#!/bin/bash
# round 6, first call: the fused predict_feature tail -- parity tests, A/B of the feature-mode configs, kernel stats
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "feat_tail or blend or grid_sample" -x > $O/pytest_ops.txt 2>&1; echo "ops exit=$?"; tail -3 $O/pytest_ops.txt
timeout -k 10 500 python -m pytest tests/test_gpu_flow.py tests/test_gpu_vit.py -q -m gpu -x -k "feature or toy" > $O/pytest_flow.txt 2>&1; echo "flow exit=$?"; tail -3 $O/pytest_flow.txt
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "feature" > $O/pytest_full.txt 2>&1; echo "full exit=$?"; tail -3 $O/pytest_full.txt
for i in 1 2; do
python tools/bench_configs.py --only feat --steps 40 2>&1 | grep -v amdgpu.ids | tail -1
python tools/bench_configs.py --only feat --steps 40 --feat-op-by-op 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/$/  [op by op]/'
python tools/bench_configs.py --only cfg3 --steps 60 2>&1 | grep -v amdgpu.ids | tail -1
python tools/bench_configs.py --only cfg3 --steps 60 --feat-op-by-op 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/$/  [op by op]/'
done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/feat -- python3 $R/tools/bench_configs.py --only feat --steps 10 > $O/feat.txt 2> $O/feat.err; echo "feat rocprof exit=$?"
f=$(find $O/feat -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $O/feat_kernel_stats.csv && head -12 "$f" | cut -c1-200
find $O -name "*kernel_trace*.csv" -delete; find $O -name "*.db" -delete
