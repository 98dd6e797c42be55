# the two-pass kernel with pass 2 back to front: parity on its tests, then 20 000 x 5 000 (driver's window and 45 steps in) and a configs[4] shard
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_pub.py tests/test_gpu_new_landmarks.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r05/t_m.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05/t_m.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 20000 --landmarks 5000 --steps 20 --warmup 5 > gpurun_out/r05/big_m.$rep.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 20000 --landmarks 5000 --steps 10 --warmup 45 > gpurun_out/r05/big_m45.$rep.json 2>/dev/null
done
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-refscene --no-probes --steps 5 --warmup 2 > gpurun_out/r05/c4_m.json 2>gpurun_out/r05/c4_m.err
python3 - <<'PY'
import json
for n in ('big_m.1','big_m.2','big_m45.1','big_m45.2'):
    d = json.load(open('gpurun_out/r05/%s.json' % n)); r = d['roofline']
    print(n, 'ms/step %.3f kernel %.3f frac %.3f route %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['route']))
d = json.load(open('gpurun_out/r05/c4_m.json'))
print(json.dumps(d.get('configs4_shard'))[:900])
PY
