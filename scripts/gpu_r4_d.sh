# round 4: which phases of k_step_pub are on the critical path?  400 extra float64 instructions per lane in one phase at a time
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_pub.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -3
AB_LIBS="libparakeet_slam.so libpk_padG0.so libpk_padK0.so libpk_padG1.so libpk_padK1.so libpk_padU0.so libpk_padU1.so" bash scripts/gpu_ab_lib.sh 2>&1 | tee gpurun_out/r04/d_pad.log
