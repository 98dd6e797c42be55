cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ml.json 2> gpurun_out/bench_ml.err; tail -3 gpurun_out/bench_ml.err; cat gpurun_out/bench_ml.json
timeout 600 python bench.py --steps 10 --warmup 3 --particles 100000 --landmarks 2000 --no-cpu-baseline > gpurun_out/bench_c3_ml.json 2> gpurun_out/bench_c3_ml.err; tail -3 gpurun_out/bench_c3_ml.err; cat gpurun_out/bench_c3_ml.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ml -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_ml.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_ml -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f"
