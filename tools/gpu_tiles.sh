#!/bin/bash
# tile sweep for every conv shape of the PSPNet window (B=2)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for shape in dec l4c2 l4c3 l4c1 l3c2 l3c3 l3c1 l2c2 l1c3 stem3; do
  for t in 1 2 3 4; do
    python3 tools/conv_bench.py $shape $t 10 2>/dev/null | tail -1
  done
done
