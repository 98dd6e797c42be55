cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -q -m gpu -k "rccl_code_path" 2>&1 | tail -5
