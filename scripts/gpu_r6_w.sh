cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pub.py -q -m gpu -k "folded or small_maps" 2>&1 | tail -5
