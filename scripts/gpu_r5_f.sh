cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for wk in "2 6" "5 10"; do set -- $wk
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --steps $2 --warmup $1 --particles 125000 --landmarks 5000 > gpurun_out/r05/bench_125k5k_w$1.json 2> gpurun_out/r05/bench_125k5k_w$1.err; echo "rc=$?"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --steps 20 --warmup 5 --particles 20000 --landmarks 5000 > gpurun_out/r05/bench_20k5k_f2.json 2> gpurun_out/r05/bench_20k5k_f2.err; echo "rc=$?"
python3 - <<'PY'
import json
for n in ('bench_125k5k_w2', 'bench_125k5k_w5', 'bench_20k5k_f2'):
    try:
        d = json.load(open('gpurun_out/r05/%s.json' % n)); r = d['roofline']
        print(n, 'ms/step %.3f kernel %.3f ms frac %.3f flagged %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('particles_sent_to_general_kernels_last_step')), d.get('summary'))
        if d.get('per_step'): print('   ', json.dumps(d['per_step'])[:600])
    except Exception as e: print(n, 'unreadable', e)
PY
