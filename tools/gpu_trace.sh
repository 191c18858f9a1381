#!/bin/bash
# Workgroup timelines of conv_igemm_dma_f32 (tools/probe_conv_trace.hip, built with -DFS_TRACE) on the layer shapes
# that matter (B=2, 90x90 maps), plus the de-phasing experiment (dbg 32 | n<<8: workgroups bid+256 start n*1024 cycles late).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/trace
run() { name=$1; shift; timeout -k 10 120 tools/bin/probe_conv_trace "$@" > gpurun_out/trace/$name.csv && python tools/analyze_conv_trace.py gpurun_out/trace/$name.csv; echo; }
run dec   2 90 90 4096 512 3 1 1 1
run l4c1  2 90 90 2048 512 1 0 1 1
run l4c3  2 90 90 512 2048 1 0 1 1
run l3c1  2 90 90 1024 256 1 0 1 1
run l3c3  2 90 90 256 1024 1 0 1 1
run l1c3  2 179 179 64 256 1 0 1 1
echo "== de-phased by 20 x 1024 cycles"
run l3c3_s20 2 90 90 256 1024 1 0 1 1 $(( 32 | (20 << 8) ))
run l4c3_s20 2 90 90 512 2048 1 0 1 1 $(( 32 | (20 << 8) ))
