# round 6: the new tests (duo, threads, balanced planner in isolation, pub / audit / fuzz / errors / grow), then the configs[4] shard with and without the duo instance
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_duo.py tests/test_gpu_threads.py "tests/test_gpu_sharded.py::test_balanced_plan_kernels_in_isolation_at_every_rank_of_worlds_up_to_eight" tests/test_gpu_pub.py tests/test_gpu_audit.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py tests/test_gpu_grow.py -q -m gpu > $O/f_tests.log 2>&1; echo "tests rc $?" | tee -a $O/f_tests.log
tail -25 $O/f_tests.log
for v in 1 0; do
ST_P=125000 ST_L=5000 ST_S=50 ST_OPTS=pub_duo=$v ST_OUT=$O/f_shard_duo$v.json timeout -k 10 500 python scripts/gpu_diag_pubstats.py > $O/f_shard_duo$v.log 2>&1
done
python3 - <<'PY'
import json, statistics as st
O='gpurun_out/r06'
a=json.load(open(O+'/f_shard_duo1.json'))['steps']; b=json.load(open(O+'/f_shard_duo0.json'))['steps']
for x,y in zip(a,b):
    if x['step'] % 4 == 0 or x['step'] < 6: print('step %2d duo inst %d %.2f ms (observe %.2f) flagged %d distinct %d | big %.2f ms (observe %.2f)'%(x['step'],x['instance'],x['ms'],x['spans']['observe'],x['flagged'],x['distinct_sources'],y['ms'],y['spans']['observe']))
for lo,hi in ((5,25),(40,50)):
    print('steps %d-%d: duo %.3f big %.3f'%(lo,hi-1,st.mean(x['ms'] for x in a[lo:hi]),st.mean(y['ms'] for y in b[lo:hi])))
PY
