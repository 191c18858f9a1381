#!/bin/bash
# A/B of explicit library options on ONE box, candidates interleaved: tools/gpu_ab_opts.sh STEPS "optA optB" "optC" ...   ("-" = defaults)
cd ${GRAFT_REPO_ROOT:-/root/repo}
STEPS=$1; shift
mkdir -p gpurun_out
for i in 1 2 3; do
  for cand in "$@"; do
    args=""; [ "$cand" != "-" ] && for o in $cand; do args="$args --hip-opt $o"; done
    tag=$(echo "$cand" | tr ' =' '__')
    timeout -k 10 300 python bench.py --steps $STEPS --warmup 20 --no-extras --no-cpu-baseline $args > gpurun_out/abo_${tag}_$i.json 2> gpurun_out/abo_err.txt || { tail -5 gpurun_out/abo_err.txt; exit 1; }
    python - <<PY
import json
j=json.load(open("gpurun_out/abo_${tag}_$i.json"))
print("$cand $i", j["value"], "FPS", j["ms_per_step"], "ms  median", j["median_ms_per_step"])
PY
  done
done
