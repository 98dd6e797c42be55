cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_assoc.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 && bash scripts/gpu_c1.sh
