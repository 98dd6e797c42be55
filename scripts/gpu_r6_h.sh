# round 6: the parked positives of k_step_pub_big -- its tests, the whole publish / subscribe suite and audits, then every step of the trajectory
# at 20 000 x 5 000 (step 0 included)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_pub.py tests/test_gpu_duo.py tests/test_gpu_audit.py tests/test_gpu_fuzz.py tests/test_gpu_config2.py -q -m gpu > $O/h_tests.log 2>&1; echo "tests rc $?" | tee -a $O/h_tests.log
tail -25 $O/h_tests.log
ST_P=20000 ST_L=5000 ST_S=30 ST_OPTS=pub_duo=0 ST_OUT=$O/h_pubstats_big.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/h_pubstats_big.log 2>&1
head -8 $O/h_pubstats_big.log | cut -c1-330
