cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 1100 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_window_a.json 2> gpurun_out/r05/bench_driver_window_a.err; echo "rc=$?"
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r05/bench_driver_window_a.json')); r = d['roofline']
print('ms/step %.3f kernel %.3f frac %.3f bound %s nodup %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['bound'], r.get('frac_no_duplicates')))
print('probe', json.dumps(r.get('no_resample_probe'))[:900])
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
c4 = d.get('configs4_shard'); print('c4', c4.get('ms_per_step'), (c4.get('roofline') or {}).get('frac'), c4.get('late_window', {}).get('ms_per_step'), (c4.get('late_window', {}).get('roofline') or {}).get('frac'), c4.get('error'))
c1 = d.get('configs1'); print('c1', c1.get('ms_per_step'), (c1.get('roofline') or {}).get('frac'))
print('refscene', json.dumps(d.get('refscene', {}).get('configs2_through_the_facade'))[:900])
print('per_step', json.dumps(d.get('per_step'))[:500])
PY
