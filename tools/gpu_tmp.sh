R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/stem_probe.py <<PY
import os, sys, torch
sys.path.insert(0, "$R")
from flood_uav_video_segmentation_amd import _lib
from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr
lib = _lib.load()
x = torch.randn(2, 3, 713, 713, device="cuda")
w = torch.randn(27, 64, device="cuda") * 0.1
sc, sh = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
out = torch.empty(2, 357, 357, 64, device="cuda")
for _ in range(20):
    check(lib.fs_stem_conv_nchw(ptr(x), ptr(w), ptr(sc), ptr(sh), ptr(out), 2, 713, 713, 64, 3, 3, 2, 1, stream_ptr()))
torch.cuda.synchronize()
PY
O=$R/gpurun_out/pmc_stem; rm -rf $O; mkdir -p $O
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  d=$O/$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $d -- python3 /tmp/stem_probe.py > $d.log 2>&1; echo "pass exit=$?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem_conv" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, sum(v) / len(v), len(v))
PY
find $O -name "*.csv" -size +1M -delete
