# round 6: after the fix of the parked positives (state word's upper half, counter per parity): its tests first, the rest only when they pass
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_pub.py -q -m gpu -x -k "parks or two_pass_instance_for_maps" > $O/i_tests0.log 2>&1; rc=$?; tail -5 $O/i_tests0.log
if grep -q "Memory access fault" $O/i_tests0.log; then echo "FAULT"; exit 1; fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1100 python -m pytest tests/test_gpu_pub.py tests/test_gpu_duo.py tests/test_gpu_audit.py tests/test_gpu_fuzz.py -q -m gpu > $O/i_tests.log 2>&1; rc=$?; echo "tests rc $rc" | tee -a $O/i_tests.log
tail -25 $O/i_tests.log
if grep -q "Memory access fault" $O/i_tests.log; then echo "FAULT"; exit 1; fi
[ $rc -eq 0 ] || exit $rc
for v in 0 1 2; do
ST_P=20000 ST_L=5000 ST_S=50 ST_OPTS=pub_duo=$v ST_OUT=$O/i_pubstats_$v.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/i_pubstats_$v.log 2>&1 || exit 1
done
python3 - <<'PY'
import json, statistics as st
O='gpurun_out/r06'
r=[json.load(open(O+'/i_pubstats_%d.json'%v))['steps'] for v in (0,1,2)]
for a,b,c in zip(*r):
    if a['step'] % 4 == 0 or a['step'] < 6: print('step %2d big %.3f (flagged %d) | duo inst %d %.3f | trio inst %d %.3f (flagged %d)'%(a['step'],a['ms'],a['flagged'],b['instance'],b['ms'],c['instance'],c['ms'],c['flagged']))
for lo,hi in ((5,25),(40,50)):
    print('steps %d-%d: big %.3f duo %.3f trio %.3f'%(lo,hi-1,*(st.mean(x['ms'] for x in rr[lo:hi]) for rr in r)))
PY
