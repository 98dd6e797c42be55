# A/B on one box: the EKF's three reciprocals by v_rcp_f64 + two Newton steps (-DPK_FAST_RECIP) against the IEEE division
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
AB_LIBS="libparakeet_slam.so libpk_fastrecip.so" AB_TAG=v_fastrecip_c2 bash scripts/gpu_ab3.sh 2>&1 | tee $O/v_ab_fastrecip_configs2.log | tail -6
AB_LIBS="libparakeet_slam.so libpk_fastrecip.so" AB_TAG=v_fastrecip_big AB_ARGS="--particles 20000 --landmarks 5000" bash scripts/gpu_ab3.sh 2>&1 | tee $O/v_ab_fastrecip_20000x5000.log | tail -6
AB_LIBS="libparakeet_slam.so libpk_fastrecip.so" AB_TAG=v_fastrecip_c1 AB_ARGS="--particles 10000 --landmarks 500" AB_STEPS=120 bash scripts/gpu_ab3.sh 2>&1 | tee $O/v_ab_fastrecip_configs1.log | tail -6
