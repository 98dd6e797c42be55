# A/B on one box: one logarithm per lane and particle for the importance factors' norms (-DPK_ONE_LOG) against one per update
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
AB_LIBS="libparakeet_slam.so libpk_onelog.so" AB_TAG=u_onelog_c2 bash scripts/gpu_ab3.sh 2>&1 | tee $O/u_ab_onelog_configs2.log | tail -6
AB_LIBS="libparakeet_slam.so libpk_onelog.so" AB_TAG=u_onelog_big AB_ARGS="--particles 20000 --landmarks 5000" bash scripts/gpu_ab3.sh 2>&1 | tee $O/u_ab_onelog_20000x5000.log | tail -6
