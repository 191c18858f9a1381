#!/bin/bash
# Power / clock evidence (run through gpurun): is the window bound by the matrix rate at the nominal clock or by the energy it takes
# at the board's power limit?  Samples hwmon while the headline workload loops, on both arithmetic routes, plus an HBM-bound
# comparison (the tail + interpolation only: the key-frame cache variant would mix both).
set -e
O=gpurun_out/power
mkdir -p $O
ls -l /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/hwmon_ls.txt 2>&1 || true
rocm-smi --showmaxpower --showpower --showclocks --showperflevel > $O/rocm_smi_idle.txt 2>&1 || true
STEPS=${1:-1500}
python3 tools/power_trace.py -- python3 tools/bench_configs.py --only cfg1 --steps $STEPS --json > $O/split_route.json
python3 tools/power_trace.py -- python3 tools/bench_configs.py --only cfg1 --steps $((STEPS / 2)) --json --opt hip_no_split_bf16 > $O/fp32_route.json
python3 tools/power_trace.py -- python3 tools/bench_configs.py --only cfg3 --steps $STEPS --json > $O/vit_s16.json
python3 tools/power_trace.py -- python3 tools/bench_configs.py --only cfg1 --steps $STEPS --json > $O/split_route_again.json
cat $O/split_route.json $O/fp32_route.json $O/vit_s16.json $O/split_route_again.json
