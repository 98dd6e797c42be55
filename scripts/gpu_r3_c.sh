# per-phase stamps of k_step_pub at 51 200 x 2 000
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ST_P=${ST_P:-51200} ST_L=2000 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r3c_stamps.txt 2>&1; echo rc=$?; cat gpurun_out/r3c_stamps.txt | tail -14
