#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
for rep in 1 2; do
for d in 0 2 3 4 6 8; do
  if [ $d == 0 ]; then opt=""; else opt="hip_res_touch=$d"; fi
  timeout -k 10 300 python tools/layer_profile.py 2 pspnet50 $opt > $O/r5c_layers_touch${d}_$rep.txt 2>&1 || exit 1
  echo "touch=$d rep=$rep $(grep -E '^total' $O/r5c_layers_touch${d}_$rep.txt) $(grep -E 'split128x128 ' $O/r5c_layers_touch${d}_$rep.txt)"
done
done
