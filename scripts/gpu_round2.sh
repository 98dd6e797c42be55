# round 2 check: whole -m gpu suite, smoke, the default bench (configs[2] + configs[1] + probes + CPU baseline), kernel trace of the
# same command, PMC traffic at configs[2], per-phase stamps of k_step_regs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -1 gpurun_out/bench_default.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_default.log 2>&1; echo "prof rc $?"
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/prof_default/*/*kernel_stats.csv | head -1); cut -c1-150 $f | head -14
bash scripts/gpu_pmc_traffic2.sh
ST_P=51200 ST_L=2000 timeout 300 python scripts/gpu_stamps.py > gpurun_out/stamps_regs.txt 2>&1; cat gpurun_out/stamps_regs.txt
