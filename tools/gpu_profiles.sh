#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries for the headline bench (configs[1]) and for configs[2] / configs[3]
# (tools/bench_configs.py --only cfgN), + the per-layer HIP-event tables.  Outputs under gpurun_out/prof_$ROUND/ (ROUND: round tag, default r03).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r06}
O=$R/gpurun_out/prof_$ROUND
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/bench.err; echo "bench rocprof exit=$?"
for c in cfg2 cfg3 crops feat; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -- python3 $R/tools/bench_configs.py --only $c --steps 10 > $O/$c.txt 2> $O/$c.err; echo "$c rocprof exit=$?"; grep configs $O/$c.txt
done
for d in bench cfg2 cfg3 crops feat; do
  f=$(find $O/$d -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv && head -8 "$f" | cut -c1-160
done
find $O -name "*kernel_trace*.csv" -delete
find $O -name "*.db" -delete
cd $R
python tools/layer_profile.py 2 2>&1 | grep -v amdgpu.ids > $O/layers_b2.txt; tail -8 $O/layers_b2.txt
python tools/layer_profile.py 2 deeplab101 2>&1 | grep -v amdgpu.ids > $O/layers_deeplab101_b2.txt; tail -9 $O/layers_deeplab101_b2.txt
python tools/vit_profile.py s16 2>&1 | grep -v amdgpu.ids > $O/vit_s16_layers.txt; tail -3 $O/vit_s16_layers.txt
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > $O/all_configs_1gpu.txt; cat $O/all_configs_1gpu.txt
