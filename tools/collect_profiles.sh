#!/bin/bash
# Copy what the last tools/gpu_round.sh left under gpurun_out/ into the tracked profiles/ directory (round tag as $1, e.g. r02).
set -e
cd "$(dirname "$0")/.."
T=${1:-r02}; O=gpurun_out/prof_r02
cp $O/bench_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
cp $O/cfg2_kernel_stats.csv profiles/${T}_cfg2_deeplabv3_r101_kernel_stats.csv
cp $O/cfg3_kernel_stats.csv profiles/${T}_cfg3_vit_s16_kernel_stats.csv
cp $O/layers_b2.txt profiles/${T}_layers_b2.txt
cp $O/layers_deeplab101_b2.txt profiles/${T}_layers_deeplab101_b2.txt
cp $O/vit_s16_layers.txt profiles/${T}_vit_s16_layers.txt
cp $O/all_configs_1gpu.txt profiles/${T}_all_configs_1gpu.txt
cp $O/bench_under_rocprof.json profiles/${T}_bench_under_rocprof.json
cp gpurun_out/pmc_bench_summary.json profiles/${T}_pmc_traffic.json
cp gpurun_out/parity_measured.txt profiles/${T}_parity_measured.txt
[ -s gpurun_out/bench_n1.json ] && cp gpurun_out/bench_n1.json profiles/${T}_bench_n1.json
ls profiles | grep "^${T}_" | tr '\n' ' '; echo
