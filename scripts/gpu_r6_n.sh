# A/B on one box at 20 000 x 5 000: the library as it is against a diagnostic variant (AB_VARIANT=libpk_<name>.so)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
V=${AB_VARIANT:-libpk_recahead.so}
AB_LIBS="libparakeet_slam.so $V" AB_TAG=n_${V%.so} AB_ARGS="--particles 20000 --landmarks 5000 --no-configs4 --no-refscene" bash scripts/gpu_ab3.sh 2>&1 | tee $O/n_ab_${V%.so}.log | tail -8
