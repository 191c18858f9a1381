#!/bin/bash
# HBM traffic of the dominant kernel during bench.py: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots),
# kernel-trace only (no sys/hip/hsa trace domains together with --pmc).  Summary -> gpurun_out/pmc_bench_summary.json
# (copy it to profiles/rNN_pmc_traffic.json: bench.py quotes it only while its build_id matches the running sources).
# usage on the GPU box: GIT_HEAD=<short sha> tools/gpu_pmc_bench.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcb
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/$(echo $C | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $d -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $d.log 2>&1; echo "pass $C exit=$?"
done
python3 - <<PY
import csv, glob, collections, json, os, sys, hashlib
sys.path.insert(0, "$R")
import bench
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if not k.startswith("void fs::conv_igemm"):
        continue
    e = {c: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
    out[k] = e
so = open("$R/flood_uav_video_segmentation_amd/libfloodseg.so", "rb").read()
meta = {"build_id": bench.build_id(), "git_head": os.environ.get("GIT_HEAD", "unknown"), "so_sha256": hashlib.sha256(so).hexdigest()[:16],
        "command": "rocprofv3 --kernel-trace --pmc <C> -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras, one pass per counter group",
        "formula": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 reports KiB; gfx950 counts a 128-B fetch as 64 B)"}
json.dump({"meta": meta, "kernels": out}, open("$R/gpurun_out/pmc_bench_summary.json", "w"), indent=1)
for k, e in out.items():
    print(k[:60], {c: (round(v["mean"], 1) if isinstance(v, dict) else round(v / 1e6, 2)) for c, v in e.items()})
PY
find $O -name "*kernel_trace.csv" -delete
