cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python ${DG_SCRIPT:-scripts/gpu_diag_pub2.py} 2>&1 | tail -30
