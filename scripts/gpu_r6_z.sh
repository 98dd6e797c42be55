# A/B on one box at configs[1]: the flag launch folded into k_cand_entries against a launch of its own (libpk_prev.so = the commit before)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
AB_LIBS="libpk_prev.so libparakeet_slam.so" AB_TAG=z_flag_fold_c1 AB_ARGS="--particles 10000 --landmarks 500" AB_STEPS=120 bash scripts/gpu_ab3.sh 2>&1 | tee $O/z_ab_flag_fold_configs1.log | tail -6
