#!/bin/bash
# Tile choice for the Segmenter's nn.Linear GEMMs at B=2 (ViT-S/16 @704: 2 x 1937 = 3874 rows; ViT-B/32 @704: 2 x 485 = 970).
cd "$(dirname "$0")/.."
for shape in "3874 384 1152" "3874 384 384" "3874 384 1536" "3874 1536 384" "970 768 2304" "970 768 768" "970 768 3072" "970 3072 768"; do
  set -- $shape
  for t in 0 1 2 4 3; do
    timeout -k 10 120 tools/bin/probe_conv_trace 1 $1 1 $2 $3 1 0 1 $t 0 | head -1
  done
done
