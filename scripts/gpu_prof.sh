cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_x
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/prof_x.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_x -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_x.csv && cut -c1-160 "$f" | head -${NLINES:-14}
