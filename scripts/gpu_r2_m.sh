# single-rank sharded code path against the plain path at the default workload (verdict item 8: within 3 %)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for mode in "" "--force-sharded"; do
timeout 400 python bench.py --no-cpu-baseline --no-secondary --no-probes --steps 30 --warmup 5 $mode 2>gpurun_out/m_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode [$mode] ms/step %.4f observe %.4f parallelism %s' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['config']['parallelism']))" || tail -3 gpurun_out/m_err.txt
done; done
for mode in "" "--force-sharded"; do
timeout 400 python bench.py --no-cpu-baseline --no-secondary --no-probes --steps 200 --warmup 10 --particles 10000 --landmarks 500 $mode 2>gpurun_out/m_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[1] mode [$mode] ms/step %.4f' % (d['ms_per_step']))" || tail -3 gpurun_out/m_err.txt
done
