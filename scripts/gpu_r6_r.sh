cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
ST_S=50 timeout -k 10 400 python scripts/gpu_diag_rounds.py > $O/r_rounds_20480x5000.log 2>&1; echo rc $?
tail -22 $O/r_rounds_20480x5000.log
