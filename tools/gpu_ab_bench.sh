#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by +-4 %): tools/gpu_ab_bench.sh <other.so> [steps]
# alternates `bench.py --lib <other.so>` and the in-tree build, three rounds, headline only.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OTHER=$1; STEPS=${2:-150}
mkdir -p gpurun_out
for i in 1 2 3; do
  for which in other tree; do
    if [ $which == other ]; then L="--lib $OTHER"; else L=""; fi
    timeout -k 10 300 python bench.py --steps $STEPS --warmup 20 --no-extras --no-cpu-baseline $L > gpurun_out/ab_${which}_$i.json 2> gpurun_out/ab_err.txt || { tail -5 gpurun_out/ab_err.txt; exit 1; }
    python - <<PY
import json
j=json.load(open("gpurun_out/ab_${which}_$i.json"))
r=j["roofline"]
print("$which $i", j["value"], "FPS", j["ms_per_step"], "ms  median", j["median_ms_per_step"], " dom", r["avg_launch_ms"], " small:", {k:v for k,v in r["per_kernel_ms_per_step"].items() if not k.startswith(("split","wino"))})
PY
  done
done
