cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_assoc.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 && bash scripts/gpu_bench_quick.sh && bash scripts/gpu_c5.sh
