#!/bin/bash
# Round 5: what do the pixel tile's trips through LDS cost the split main loop?  Elimination runs of the FS_TRACE probe (results invalid,
# timing only): dbg 64 = no pixel-row DMAs, 128 = no pixel-fragment LDS reads, 192 = neither, 256 = no filter-row DMAs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export FS_WARM=50 FS_AVG=200
for shape in "2 90 90 2048 512 1 0 1 1" "2 90 90 512 2048 1 0 1 1" "2 90 90 256 1024 1 0 1 1" "2 90 90 1024 256 1 0 1 2"; do
  echo "== $shape"
  for rep in 1 2; do
    for dbg in 0 64 128 192 256; do
      tools/bin/probe_conv_trace $shape $dbg 1 0 1 | tail -1
    done
  done
done
