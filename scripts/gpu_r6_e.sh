# round 6: per-wave stamps of the two instances of the two-pass kernel (20 480 x 5 000, 15 steps in), then the rest of the new tests
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
ST_P=20480 ST_L=5000 ST_WARM=15 timeout -k 10 300 python scripts/gpu_stamps.py > $O/e_stamps_duo.txt 2>&1
PK_OPT_PUB_DUO=0 ST_P=20480 ST_L=5000 ST_WARM=15 timeout -k 10 300 python scripts/gpu_stamps.py > $O/e_stamps_big.txt 2>&1
cat $O/e_stamps_duo.txt $O/e_stamps_big.txt
timeout -k 10 1100 python -m pytest tests/test_gpu_duo.py tests/test_gpu_threads.py "tests/test_gpu_sharded.py::test_balanced_plan_kernels_in_isolation_at_every_rank_of_worlds_up_to_eight" tests/test_gpu_pub.py tests/test_gpu_audit.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py tests/test_gpu_grow.py -x -q -m gpu > $O/e_tests.log 2>&1; echo "tests rc $?" | tee -a $O/e_tests.log
tail -15 $O/e_tests.log
