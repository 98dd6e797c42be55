# full GPU check of a round: tests, smoke, bench, rocprof summary
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
timeout 600 python bench.py --assoc known --no-cpu-baseline --steps 50 > gpurun_out/bench_known.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --particles 100000 --landmarks 2000 > gpurun_out/bench_c3_ml.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --particles 100000 --landmarks 2000 --assoc known > gpurun_out/bench_c3_known.json 2>/dev/null
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_default
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_default.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_default -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_default.csv && head -6 "$f"
# config 3 kernel trace (ML route: k_assoc_grid hand-off + k_observe_sweep)
bash scripts/gpu_prof_c3.sh > gpurun_out/prof_c3_head.txt 2>&1; tail -8 gpurun_out/prof_c3_head.txt | cut -c1-160
