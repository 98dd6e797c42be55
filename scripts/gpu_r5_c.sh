# round 5: the scan-level far pruning -- parity suites of the one-pass routes, then A/B bench lines (far_prune 1 / 0)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_pub.py tests/test_gpu_audit.py tests/test_gpu_fuzz.py tests/test_gpu_random_worlds.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05/t_pub.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05/t_pub.log
for fp in 1 0; do
PK_OPT_FAR_PRUNE=$fp timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --steps 20 --warmup 5 > gpurun_out/r05/bench_c2_fp$fp.json 2> gpurun_out/r05/bench_c2_fp$fp.err; echo "bench c2 fp=$fp rc=$?"
PK_OPT_FAR_PRUNE=$fp timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --steps 20 --warmup 5 --particles 20000 --landmarks 5000 > gpurun_out/r05/bench_20k5k_fp$fp.json 2> gpurun_out/r05/bench_20k5k_fp$fp.err; echo "bench 20k5k fp=$fp rc=$?"
done
python3 - <<'PY'
import json
for n in ('bench_c2_fp1', 'bench_c2_fp0', 'bench_20k5k_fp1', 'bench_20k5k_fp0'):
    try:
        d = json.load(open('gpurun_out/r05/%s.json' % n)); r = d['roofline']
        print(n, 'ms/step %.3f kernel %.3f ms frac %.3f no-dup %s flagged %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('frac_no_duplicates'), r.get('particles_sent_to_general_kernels_last_step')), d.get('summary'))
        if d.get('per_step'): print('   ', json.dumps(d['per_step'])[:400])
    except Exception as e: print(n, 'unreadable', e)
PY
