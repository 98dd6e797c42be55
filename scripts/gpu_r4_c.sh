# round 4: selected test files, then an A/B of library builds (AB_LIBS) and the slow-window replay of the default bench
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest ${R4_TESTS:-tests/test_gpu_pub.py tests/test_gpu_audit.py tests/test_gpu_config2.py} -m gpu -x -q --no-header -p no:cacheprovider > gpurun_out/r04/c_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r04/c_tests.log
[ -n "$AB_LIBS" ] && bash scripts/gpu_ab_lib.sh 2>&1 | tee gpurun_out/r04/c_ab.log
[ -n "$R4_BENCH" ] && (timeout -k 10 500 python bench.py $R4_BENCH > gpurun_out/r04/c_bench.json 2> gpurun_out/r04/c_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/c_bench.json'))
r = d['roofline']
print('ms/step %.3f value %.4g frac %.4f bound %s issue %s' % (d['ms_per_step'], d['value'], r['frac'], r['bound'], (r.get('issue') or {}).get('frac')))
print('per_step', json.dumps(d.get('per_step'))[:600])
for k in ('configs1', 'configs4_shard'):
    if k in d: print(k, d[k].get('ms_per_step'), (d[k].get('roofline') or {}).get('frac'), d[k].get('error'))
print('refscene', json.dumps(d.get('refscene'))[:700])
PY
)
exit $rc
