# k_step_pub iteration: ML test files, then the kernel trace of the default bench (only the timed filter)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_assoc.py tests/test_gpu_random_worlds.py tests/test_gpu_new_landmarks.py tests/test_gpu_config2.py -x -q -m gpu > gpurun_out/r3d_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3d_tests.txt
bash scripts/gpu_r3_b.sh 2>&1 | head -4 | cut -c1-200
