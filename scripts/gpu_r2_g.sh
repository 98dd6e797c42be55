cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_assoc.py tests/test_gpu_random_worlds.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/pytest_g.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/pytest_g.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/pytest_g.log; exit 1; fi
AB_LIBS="libpk_a.so libparakeet_slam.so" bash scripts/gpu_ab_lib.sh
