cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_pub.py tests/test_gpu_duo.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -4
