#!/bin/bash
# Tile choice for the grouped Winograd GEMMs (36 groups): rows per group, Cin, Cout as the network launches them at B=2.
cd "$(dirname "$0")/.."
for shape in "1152 256 256" "1152 512 512" "1058 2048 512"; do
  set -- $shape
  for t in 1 2 4 3; do
    timeout -k 10 120 tools/bin/probe_conv_trace 1 $1 1 $2 $3 1 0 1 $t 0 36 | head -1
  done
done
