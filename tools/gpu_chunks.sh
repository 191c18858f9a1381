#!/bin/bash
# Round 5: per-chunk durations of the split main loop (FS_TRACE probe, scalar stamps at the end of each of the first 8 K chunks)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export FS_WARM=100 FS_CHUNKS=1
echo "== layer3 conv3 K=256 N=1024 +res (1016 tiles, 8 chunks)";  tools/bin/probe_conv_trace 2 90 90 256 1024 1 0 1 1 0 1 1 1
echo "== layer3 wino GEMM 64 groups K=256 N=256 (512 tiles)";      tools/bin/probe_conv_trace 1 512 1 256 256 1 0 1 1 0 64 0 1
echo "== layer3 conv1 K=1024 N=256 128x64 (508 tiles, first 8 of 32)"; tools/bin/probe_conv_trace 2 90 90 1024 256 1 0 1 2 0 1 0 1
echo "== layer4 conv3 K=512 N=2048 +res (2032 tiles, first 8 of 16)"; tools/bin/probe_conv_trace 2 90 90 512 2048 1 0 1 1 0 1 1 1
echo "== layer4 conv1 K=2048 N=512 (508 tiles, first 8 of 64)";    tools/bin/probe_conv_trace 2 90 90 2048 512 1 0 1 1 0 1 0 1
