#!/bin/bash
# Copy what the last tools/gpu_round.sh left under gpurun_out/ into the tracked profiles/ directory (round tag as $1, e.g. r02).
set -e
cd "$(dirname "$0")/.."
T=${1:-r06}; O=gpurun_out/prof_$T
[ -d "$O" ] || { echo "no $O: run ROUND=$T tools/gpu_round.sh first (this script never copies another round's artefacts)"; exit 1; }
cp $O/bench_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
cp $O/cfg2_kernel_stats.csv profiles/${T}_cfg2_deeplabv3_r101_kernel_stats.csv
cp $O/cfg3_kernel_stats.csv profiles/${T}_cfg3_vit_s16_kernel_stats.csv
[ -s $O/crops_kernel_stats.csv ] && cp $O/crops_kernel_stats.csv profiles/${T}_real_video_route_8crops_kernel_stats.csv
[ -s $O/feat_kernel_stats.csv ] && cp $O/feat_kernel_stats.csv profiles/${T}_feature_mode_kernel_stats.csv
cp $O/layers_b2.txt profiles/${T}_layers_b2.txt
cp $O/layers_deeplab101_b2.txt profiles/${T}_layers_deeplab101_b2.txt
cp $O/vit_s16_layers.txt profiles/${T}_vit_s16_layers.txt
cp $O/all_configs_1gpu.txt profiles/${T}_all_configs_1gpu.txt
cp $O/bench_under_rocprof.json profiles/${T}_bench_under_rocprof.json
# the PMC summary records the build it was measured on: refuse one that is not the sources in this tree
python3 - <<PY || exit 1
import json, sys
sys.path.insert(0, ".")
import bench
m = json.load(open("gpurun_out/pmc_bench_summary.json")).get("meta", {})
if m.get("build_id") != bench.build_id():
    sys.exit(f"gpurun_out/pmc_bench_summary.json was measured on build {m.get('build_id')}, the tree is {bench.build_id()}: stale, not copied")
PY
cp gpurun_out/pmc_bench_summary.json profiles/${T}_pmc_traffic.json
cp gpurun_out/parity_measured.txt profiles/${T}_parity_measured.txt
[ -s gpurun_out/bench_n1.json ] && cp gpurun_out/bench_n1.json profiles/${T}_bench_n1.json
ls profiles | grep "^${T}_" | tr '\n' ' '; echo
