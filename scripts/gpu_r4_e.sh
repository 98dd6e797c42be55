# round 4: stamps (per phase and per wave) of k_step_pub, then an A/B with three repetitions
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
ST_P=51200 ST_L=2000 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r04/stamps_k_step_pub_51200x2000.txt 2>&1; cat gpurun_out/r04/stamps_k_step_pub_51200x2000.txt | tail -24
[ -n "$AB_LIBS" ] && (bash scripts/gpu_ab_lib.sh; AB_LIBS="$AB_LIBS" bash scripts/gpu_ab_lib.sh | head -${AB_HEAD:-2}) 2>&1 | tee gpurun_out/r04/e_ab.log
