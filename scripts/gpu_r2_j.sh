# rehearsal of bench.py's self-launched N > 1 path on ONE GPU (two ranks share cuda:0, gloo), then the sharded tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --particles 8192 --landmarks 600 --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "rc $?"; tail -3 gpurun_out/bench2.err; cut -c1-900 gpurun_out/bench2.json
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --particles 20000 --landmarks 2000 --no-cpu-baseline > gpurun_out/bench2b.json 2> gpurun_out/bench2b.err; echo "rc $?"; tail -3 gpurun_out/bench2b.err; cut -c1-700 gpurun_out/bench2b.json
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
