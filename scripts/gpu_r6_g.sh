# round 6: the sharded suite (RCCL one rank with the balanced loopback), the duo tests again
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_duo.py -q -m gpu -x > $O/g_tests.log 2>&1; echo "tests rc $?" | tee -a $O/g_tests.log
tail -30 $O/g_tests.log
