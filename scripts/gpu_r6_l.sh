# round 6: grow / facade / sharded-grow tests again; configs[1] with and without the 256-lane publish / subscribe instance (its static LDS back under a third of a CU)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_new_landmarks.py tests/test_gpu_facade.py tests/test_gpu_duo.py -q -m gpu > $O/l_tests.log 2>&1; rc=$?; echo "tests rc $rc" | tee -a $O/l_tests.log
tail -12 $O/l_tests.log
if grep -q "Memory access fault" $O/l_tests.log; then echo "FAULT"; exit 1; fi
for rep in 1 2 3; do for v in 0 1; do PK_OPT_PUB_SMALL=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 120 --warmup 10 --particles 10000 --landmarks 500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pub_small $v ms/step %.4f observe %.4f route %s' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route']))"; done; done | tee $O/l_ab_pub_small.log
for v in 0 2; do ST_P=20000 ST_L=5000 ST_S=30 ST_OPTS=pub_duo=$v ST_OUT=$O/l_pubstats_$v.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/l_pubstats_$v.log 2>&1; done
python3 - <<'PY'
import json, statistics as st
O='gpurun_out/r06'
r=[json.load(open(O+'/l_pubstats_%d.json'%v))['steps'] for v in (0,2)]
print('steps 5-24: big %.3f trio %.3f'%tuple(st.mean(x['ms'] for x in rr[5:25]) for rr in r))
PY
