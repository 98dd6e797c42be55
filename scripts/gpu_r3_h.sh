# rehearsal of bench.py --gpus 4 at configs[3]'s full size on ONE GPU (four ranks of 100 000 x 2 000, gloo): control flow,
# migration volume per step; the time is of no interest (four processes share one device)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 4 --steps ${N4_STEPS:-30} --warmup 5 --no-cpu-baseline --no-probes > gpurun_out/r3h_bench4.json 2> gpurun_out/r3h_err.txt; echo rc=$?; tail -3 gpurun_out/r3h_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3h_bench4.json'))
print({k: d.get(k) for k in ('n_gpus','ms_per_step','value','migrated_particles_per_step','migrated_bytes_per_step')}); print(d['config']['parallelism'])
PY
