#!/bin/bash
# Short-K 1x1 layers: where does the time go?  dbg16 = skip the epilogue, dbg2 = one workgroup per CU.
cd "$(dirname "$0")/.."
for s in l3c3 l3c1 l4c3 l4c1 l1c3; do
  for t in 1 2 4 3; do
    for d in 0 16 2; do
      python tools/conv_bench.py $s $(( t | (d << 11) )) 30 2>/dev/null | grep -v amdgpu.ids
    done
  done
done
