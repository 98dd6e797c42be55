# first GPU check of k_observe_sweep: association tests, then config 3 ML bench (default route = sweep)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_assoc.py -x -q -m gpu 2>&1 | tail -15 &&
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --particles 100000 --landmarks 2000 > gpurun_out/bench_c3_ml_sweep.json 2> gpurun_out/bench_c3_ml_sweep.err; tail -2 gpurun_out/bench_c3_ml_sweep.err; cat gpurun_out/bench_c3_ml_sweep.json
