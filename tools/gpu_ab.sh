#!/bin/bash
# A/B of conv kernel variants (tile bit 8 = bring-up structure, bit 10 = chunk-major k order), interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for shape in dec l4c2 l4c3 l4c1 l3c2 l3c3 l3c1 l2c2 l1c3 stem3; do
  for t in ${TILES:-0x001 0x101 0x401}; do
    python3 tools/conv_bench.py $shape $t 10 2>/dev/null | tail -1
  done
done
done
