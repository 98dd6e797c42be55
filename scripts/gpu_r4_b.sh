# round 4: the tail-lane regression test against the diagnostic build with round 3's indexing (expected to FAIL), the whole -m gpu
# suite at HEAD, then an A/B of the kernel variants (AB_LIBS) at the driver's bench window
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
PK_TEST_LIB=libpk_prev.so timeout -k 10 300 python -m pytest tests/test_gpu_pub.py -q -k lanes_beyond --no-header -p no:cacheprovider 2>&1 | tail -18 > gpurun_out/r04/b_tail_prev.log; tail -16 gpurun_out/r04/b_tail_prev.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider > gpurun_out/r04/b_gpu_tests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -15 gpurun_out/r04/b_gpu_tests.log
[ -n "$AB_LIBS" ] && bash scripts/gpu_ab_lib.sh 2>&1 | tee gpurun_out/r04/b_ab.log
