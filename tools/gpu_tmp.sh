R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_flow.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 || exit 1
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_tmp; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; echo "exit=$?"
f=$(find $O/bench -name "*kernel_stats*.csv" | head -1); cp "$f" $R/gpurun_out/tmp_kernel_stats.csv
find $O -name "*.db" -delete; find $O -name "*kernel_trace*.csv" -delete
