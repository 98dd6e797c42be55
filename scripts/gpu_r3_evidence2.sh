# round-3 evidence, second part (same commit as gpu_r3_evidence.sh; PK_GIT_SHA passed in):
#   5. the L > 2 048 route at 20 000 x 5 000: bench line, PMC traffic   6. the default bench line (50 steps, CPU baseline, per-step replay)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r03
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --particles 20000 --landmarks 5000 > gpurun_out/r03/bench_20000x5000_ml.json 2> gpurun_out/r03/bench_20000x5000_err.txt; echo "bench 20000x5000 rc=$?"
PMC_P=20000 PMC_L=5000 bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_20000x5000.json gpurun_out/r03/
cd $R
timeout -k 10 900 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default_err.txt; echo "bench default rc=$?"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_driver_window.json 2> gpurun_out/r03/bench_driver_err.txt; echo "bench driver-window rc=$?"
python3 - <<'PY'
import json
for n in ('bench_20000x5000_ml', 'bench_default', 'bench_driver_window'):
    try:
        d = json.load(open('gpurun_out/r03/%s.json' % n)); r = d['roofline']
        print(n, 'ms/step %.3f value %.4g kernel %.3f ms frac %.3f no-dup %s traffic %s' % (d['ms_per_step'], d['value'], r['avg_launch_ms'], r['frac'], r.get('frac_no_duplicates'), r.get('traffic')))
        if d.get('per_step'): print('   per_step', json.dumps(d['per_step'])[:600])
        if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline'])
        if d.get('configs1'): print('   configs1', d['configs1']['ms_per_step'], d['configs1']['roofline']['frac'])
    except Exception as e:
        print(n, 'unreadable', e)
PY
