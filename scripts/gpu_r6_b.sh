# round 6: first run of k_step_pub_duo -- its own tests, the publish / subscribe suite, the audits and fuzzers; then the step times with and
# without it at 20 000 x 5 000
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_duo.py -x -q -m gpu > $O/b_duo_tests.log 2>&1; echo "duo tests rc $?" | tee -a $O/b_duo_tests.log
tail -15 $O/b_duo_tests.log
ST_P=20000 ST_L=5000 ST_S=50 ST_OUT=$O/b_pubstats_duo.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/b_pubstats_duo.log 2>&1
ST_P=20000 ST_L=5000 ST_S=50 ST_OPTS=pub_duo=0 ST_OUT=$O/b_pubstats_big.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/b_pubstats_big.log 2>&1
python3 - <<'PY'
import json
O='gpurun_out/r06'
a=json.load(open(O+'/b_pubstats_duo.json'))['steps']; b=json.load(open(O+'/b_pubstats_big.json'))['steps']
for x,y in zip(a,b):
    print('step %2d duo inst %d %.3f ms (observe %.3f) flagged %d | big %.3f ms (observe %.3f) flagged %d'%(x['step'],x['instance'],x['ms'],x['spans']['observe'],x['flagged'],y['ms'],y['spans']['observe'],y['flagged']))
import statistics as st
for lo,hi in ((5,25),(40,50)):
    print('steps %d-%d: duo %.3f big %.3f'%(lo,hi-1,st.mean(x['ms'] for x in a[lo:hi]),st.mean(y['ms'] for y in b[lo:hi])))
PY
