#!/bin/bash
# Every conv GEMM shape of PSPNet-R50 at B=2, 713x713 (stride-1 ones) x the four tile shapes + the heuristic's pick (tile 0).
cd "$(dirname "$0")/.."
P=tools/bin/probe_conv_trace
run() { # name count args...
  name=$1; cnt=$2; shift 2
  line="$name x$cnt:"
  for t in 0 1 2 4 3; do
    r=$(timeout -k 10 60 $P "$@" $t 0 ${G:-1} | head -1 | sed -E 's/^# (igemm[0-9x]+) .* ([0-9.]+) ms .*/\1 \2/')
    line="$line  t$t=$r"
  done
  echo "$line"
}
run l0.3      1 2 357 357 64 64 3 1 1
run l0.6      1 2 357 357 64 128 3 1 1
run l1.0.c1   1 2 179 179 128 64 1 0 1
run l1.c2     3 2 179 179 64 64 3 1 1
run l1.0.ds   1 2 179 179 128 256 1 0 1
run l1.c3     3 2 179 179 64 256 1 0 1
run l1.c1     2 2 179 179 256 64 1 0 1
run l2.0.c1   1 2 179 179 256 128 1 0 1
run l2.c3     4 2 90 90 128 512 1 0 1
run l2.c1     3 2 90 90 512 128 1 0 1
run l2.c2     3 2 90 90 128 128 3 1 1
run l3.0.c1   1 2 90 90 512 256 1 0 1
run l3.0.ds   1 2 90 90 512 1024 1 0 1
run l3.c3     6 2 90 90 256 1024 1 0 1
run l3.c1     5 2 90 90 1024 256 1 0 1
run l4.0.c1   1 2 90 90 1024 512 1 0 1
run l4.0.ds   1 2 90 90 1024 2048 1 0 1
run l4.c3     3 2 90 90 512 2048 1 0 1
run l4.c1     2 2 90 90 2048 512 1 0 1
G=36 run l3.wino   6 1 1152 1 256 256 1 0 1
G=36 run l4.wino   3 1 1152 1 512 512 1 0 1
G=36 run dec.wino  1 1 1058 1 2048 512 1 0 1
