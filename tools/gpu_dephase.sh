#!/bin/bash
# Round 5: start the second workgroup of every CU a fraction of a K chunk late (FS_TRACE probe, dbg = 32 | n << 8: n x 256 cycles) -- does
# de-phasing two co-resident workgroups by LESS than a chunk help the short-K launches whose workgroups run in lock-step?
cd ${GRAFT_REPO_ROOT:-/root/repo}
export FS_WARM=50 FS_AVG=200
for rep in 1 2; do
for n in 0 2 4 6 8 12 16; do
  dbg=$(( n == 0 ? 0 : 32 + n * 256 ))
  echo -n "l3c3 K=256 N=1024 +res  n=$n: "; tools/bin/probe_conv_trace 2 90 90 256 1024 1 0 1 1 $dbg 1 1 1 | tail -1
  echo -n "l3c1 K=1024 N=256 (128x64) n=$n: "; tools/bin/probe_conv_trace 2 90 90 1024 256 1 0 1 2 $dbg 1 0 1 | tail -1
  echo -n "l3 wino gemm 64 groups K=256 n=$n: "; tools/bin/probe_conv_trace 1 512 1 256 256 1 0 1 1 $dbg 64 0 1 | tail -1
done; done
