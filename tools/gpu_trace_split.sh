#!/bin/bash
# Workgroup timelines of the split-operand kernel as shipped (4 x 1 waves), FS_TRACE build of tools/probe_conv_trace.hip:
# per launch the prologue / main loop / epilogue cycles and the shader clock the card holds INSIDE that launch (s_memtime cycles
# per wall ns), after 20 and after 400 back-to-back launches of the same conv.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/trace
run() { name=$1; shift; timeout -k 10 120 tools/bin/probe_conv_trace "$@" > gpurun_out/trace/$name.csv && python tools/analyze_conv_trace.py gpurun_out/trace/$name.csv; echo; }
for warm in 20 400; do
    export FS_WARM=$warm
    echo "===== $warm launches before the traced one"
    #            B  H  W  Cin  Cout K pad dil tile dbg groups residual split
    run l4c1_$warm 2 90 90 2048 512  1 0 1 1 0 1 0 1
    run l4c3_$warm 2 90 90 512  2048 1 0 1 1 0 1 1 1
    run l3c1_$warm 2 90 90 1024 256  1 0 1 2 0 1 0 1
    run l3c3_$warm 2 90 90 256  1024 1 0 1 1 0 1 1 1
    run l1c3_$warm 2 179 179 64 256  1 0 1 1 0 1 1 1
done
