cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_pub.py tests/test_gpu_fullsize.py tests/test_gpu_random_worlds.py tests/test_gpu_config2.py -q -m gpu > $O/q_tests.log 2>&1; rc=$?; echo "tests rc $rc" | tee -a $O/q_tests.log
tail -8 $O/q_tests.log
if grep -q "Memory access fault" $O/q_tests.log; then echo "FAULT"; exit 1; fi
