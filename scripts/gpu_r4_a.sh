# round 4, first call: (1) the tail-lane regression test against the round-3 build (expected to FAIL: shows the test bites),
# (2) the whole -m gpu suite at HEAD, (3) the driver's bench line
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
PK_TEST_LIB=libpk_prev.so timeout -k 10 300 python -m pytest tests/test_gpu_pub.py -q -k lanes_beyond -x --no-header -p no:cacheprovider > gpurun_out/r04/a_tail_prev.log 2>&1; echo "prev build rc=$? (non-zero expected)"; tail -5 gpurun_out/r04/a_tail_prev.log
PK_TEST_LIB=libpk_prev.so timeout -k 10 300 python -m pytest tests/test_gpu_pub.py -q -k lanes_beyond --no-header -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04/a_tail_prev_all.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider > gpurun_out/r04/a_gpu_tests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -15 gpurun_out/r04/a_gpu_tests.log
[ $rc -eq 0 ] && timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/a_bench.json 2> gpurun_out/r04/a_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r04/a_bench.json
