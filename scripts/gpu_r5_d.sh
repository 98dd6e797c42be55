cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_pub.py -x -q -m gpu > gpurun_out/r05/t_pub2.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r05/t_pub2.log
for rep in 1 2; do
for lib in libparakeet_slam.so libpk_f2.so libpk_f15.so; do
PK_BENCH_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --steps 20 --warmup 5 > gpurun_out/r05/ab_$lib.$rep.json 2> gpurun_out/r05/ab_$lib.$rep.err; echo "bench $lib rc=$?"
done
done
python3 - <<'PY'
import json
for rep in (1, 2):
  for lib in ('libparakeet_slam.so', 'libpk_f2.so', 'libpk_f15.so'):
    try:
        d = json.load(open('gpurun_out/r05/ab_%s.%d.json' % (lib, rep))); r = d['roofline']
        print(lib, rep, 'ms/step %.3f kernel %.3f ms frac %.3f no-dup %.3f flagged %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('frac_no_duplicates'), r.get('particles_sent_to_general_kernels_last_step')), d['per_step']['timed_window']['ms_mean'], d['per_step']['slow_window']['ms_mean'], d['per_step']['timed_window']['flagged_particles_max'], d['per_step']['slow_window']['flagged_particles_max'])
    except Exception as e: print(lib, 'unreadable', e)
PY
