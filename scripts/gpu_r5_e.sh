cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
ST_WARM=15 ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r05/stamps_big_20480x5000.txt 2>&1; echo "rc=$?"
ST_WARM=15 ST_P=51200 ST_L=2000 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r05/stamps_pub_51200x2000.txt 2>&1; echo "rc=$?"
cat gpurun_out/r05/stamps_big_20480x5000.txt gpurun_out/r05/stamps_pub_51200x2000.txt
