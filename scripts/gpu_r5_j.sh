cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_pub.py tests/test_gpu_parity.py tests/test_gpu_facade.py tests/test_gpu_fuzz.py tests/test_gpu_assoc.py -x -q -m gpu > gpurun_out/r05/t_j.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05/t_j.log
for rep in 1 2; do for ps in 0 1; do
PK_OPT_PUB_SMALL=$ps timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 10000 --landmarks 500 --steps 120 --warmup 10 > gpurun_out/r05/c1_ps$ps.$rep.json 2>/dev/null; echo "rc=$?"
done; done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --steps 20 --warmup 5 > gpurun_out/r05/c2_j.json 2>/dev/null; echo "rc=$?"
python3 - <<'PY'
import json
for rep in (1,2):
  for ps in (0,1):
    d = json.load(open('gpurun_out/r05/c1_ps%d.%d.json' % (ps, rep))); r = d['roofline']
    print('pub_small', ps, 'ms/step %.4f kernel %.4f frac %.3f route %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['route']), d['summary'])
d = json.load(open('gpurun_out/r05/c2_j.json')); r = d['roofline']
print('c2 ms/step %.4f kernel %.4f frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']), d['summary'])
PY
