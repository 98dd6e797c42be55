# the default bench line (no CPU baseline), pretty-printed essentials
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/r3g_bench.json 2> gpurun_out/r3g_bench_err.txt; echo "bench rc=$?"; tail -3 gpurun_out/r3g_bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3g_bench.json'))
print('ms/step', d['ms_per_step'], 'value %.4g' % d['value'], 'kernel_ms', d.get('kernel_ms_per_step'))
r=d['roofline']; print(r['route'], r['kernel'][:40], 'frac', r['frac'], 'avg ms', r['avg_launch_ms'], 'no-dup', r.get('frac_no_duplicates'), 'unique', r.get('frac_unique'))
print('ekf_stage', {k: r.get('ekf_stage', {}).get(k) for k in ('ms_per_step', 'frac', 'frac_no_duplicates')})
print({k:r.get(k) for k in ('particles_sent_to_general_kernels_last_step','candidate_list_overflows_last_step')})
print('per_step', json.dumps(d.get('per_step')))
c=d.get('configs1'); 
if c: print('configs1', c['ms_per_step'], c['roofline']['route'], c['roofline']['frac'])
PY
