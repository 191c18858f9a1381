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
SQSET="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$SQSET"; do
  d=$O/$(echo $C | tr ' ' '_' | cut -c1-40)
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
    if not (k.startswith("void fs::conv_igemm") or "wino4_" in k):  # the matrix-core kernels: implicit GEMM and the one-kernel Winograd
        continue
    e = {c: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}
    if "TCC_HIT_sum" in e and e["TCC_HIT_sum"]["sum"] + e.get("TCC_MISS_sum", {"sum": 0})["sum"] > 0:
        e["l2_hit_rate"] = e["TCC_HIT_sum"]["sum"] / (e["TCC_HIT_sum"]["sum"] + e["TCC_MISS_sum"]["sum"])
    if "TCP_TCC_READ_REQ_sum" in e and e.get("GRBM_GUI_ACTIVE", {}).get("mean", 0) > 0:
        # read requests the CUs' L1s send to the XCD L2s (64 B each): the L2 -> CU stream, per shader clock and CU
        e["l2_read_bytes_per_launch"] = e["TCP_TCC_READ_REQ_sum"]["mean"] * 64.0
        e["l2_read_bytes_per_clk_per_cu"] = e["l2_read_bytes_per_launch"] / (e["GRBM_GUI_ACTIVE"]["mean"] / 8.0) / 256.0
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
    if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]["sum"] > 0:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves and are disjoint shares of a wave's
        # life (MI355X_MICROARCH.md, rocprofv3 PMC slots); SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per v_mfma_f32_32x32x2_f32)
        wc = e["SQ_WAVE_CYCLES"]["sum"]
        e["wave_time_shares"] = {"parked_waitcnt_or_barrier": e["SQ_WAIT_ANY"]["sum"] / wc, "issue_stall": e["SQ_WAIT_INST_ANY"]["sum"] / wc,
                                 "issuing": e["SQ_ACTIVE_INST_ANY"]["sum"] / wc}
        if e.get("SQ_LDS_IDX_ACTIVE", {}).get("sum", 0) > 0:
            e["lds_bank_conflict_share_of_lds_cycles"] = e["SQ_LDS_BANK_CONFLICT"]["sum"] / e["SQ_LDS_IDX_ACTIVE"]["sum"]
        e["mfma_busy_cycles_per_launch"] = e["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"]
        if e.get("GRBM_GUI_ACTIVE", {}).get("mean", 0) > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (293 k shader cycles x 8 for a 122-us launch); 1024 SIMDs share the busy count
            e["mfma_utilisation"] = e["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (e["GRBM_GUI_ACTIVE"]["mean"] / 8.0 * 1024.0)
    out[k] = e
so = open("$R/flood_uav_video_segmentation_amd/libfloodseg.so", "rb").read()
meta = {"build_id": bench.build_id(), "git_head": os.environ.get("GIT_HEAD", "unknown"), "so_sha256": hashlib.sha256(so).hexdigest()[:16],
        "command": "rocprofv3 --kernel-trace --pmc <C> -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras, one pass per counter group",
        "formula": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 reports KiB; gfx950 counts a 128-B fetch as 64 B); "
                   "mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); wave_time_shares = SQ_WAIT_ANY | SQ_WAIT_INST_ANY | "
                   "SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES"}
json.dump({"meta": meta, "kernels": out}, open("$R/gpurun_out/pmc_bench_summary.json", "w"), indent=1)
for k, e in out.items():
    print(k[:60], {c: (round(v["mean"], 1) if isinstance(v, dict) and "mean" in v else v) for c, v in e.items() if c in ("hbm_bytes_per_launch", "wave_time_shares", "lds_bank_conflict_share_of_lds_cycles", "mfma_busy_cycles_per_launch", "mfma_utilisation", "l2_hit_rate", "l2_read_bytes_per_clk_per_cu")})
PY
find $O -name "*kernel_trace.csv" -delete
