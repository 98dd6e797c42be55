# L > 2048 route (configs[4] map shape, a slice of the particles): kernel split
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 500 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --particles 20000 --landmarks 5000 > gpurun_out/n_bench_5000.json 2> gpurun_out/n_err.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/n_bench_5000.json'))
print('ms/step', d['ms_per_step'], 'kernel_ms', d['kernel_ms_per_step'])
r=d['roofline']; print(r['route'], r['frac'], r['avg_launch_ms'], 'assoc', r.get('assoc_kernel_ms'), 'ekf', r.get('ekf_stage',{}).get('avg_launch_ms'))
print({k:r.get(k) for k in ('particles_sent_to_general_kernels_last_step','candidate_list_overflows_last_step')})
PY
