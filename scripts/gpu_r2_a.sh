# round 2, first GPU call: whole -m gpu suite, the default bench (configs[2]) with every probe, kernel trace of the
# same command, PMC traffic at configs[2]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -2 gpurun_out/bench_default.err; cut -c1-3000 gpurun_out/bench_default.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-probes --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_default.log 2>&1; echo "prof rc $?"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_default/*/*kernel_stats.csv | head -1); head -12 $f
bash scripts/gpu_pmc_traffic2.sh
