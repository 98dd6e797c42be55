# round 6: the primary-blob table in both two-pass kernels -- the duo tests, the publish / subscribe suite, audits, fuzzers; then step times with
# and without the duo instance at 20 000 x 5 000
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_duo.py tests/test_gpu_pub.py tests/test_gpu_audit.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/d_tests.log 2>&1; echo "tests rc $?" | tee -a $O/d_tests.log
tail -15 $O/d_tests.log
ST_P=20000 ST_L=5000 ST_S=50 ST_OUT=$O/d_pubstats_duo.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/d_pubstats_duo.log 2>&1
ST_P=20000 ST_L=5000 ST_S=50 ST_OPTS=pub_duo=0 ST_OUT=$O/d_pubstats_big.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/d_pubstats_big.log 2>&1
python3 - <<'PY'
import json
O='gpurun_out/r06'
a=json.load(open(O+'/d_pubstats_duo.json'))['steps']; b=json.load(open(O+'/d_pubstats_big.json'))['steps']
for x,y in zip(a,b):
    if x['step'] % 3 == 0 or x['step'] < 6: print('step %2d duo inst %d %.3f ms (observe %.3f) flagged %d | big %.3f ms (observe %.3f) flagged %d'%(x['step'],x['instance'],x['ms'],x['spans']['observe'],x['flagged'],y['ms'],y['spans']['observe'],y['flagged']))
import statistics as st
for lo,hi in ((5,25),(40,50)):
    print('steps %d-%d: duo %.3f big %.3f'%(lo,hi-1,st.mean(x['ms'] for x in a[lo:hi]),st.mean(y['ms'] for y in b[lo:hi])))
PY
