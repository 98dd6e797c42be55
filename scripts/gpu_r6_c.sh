# round 6: what the vector-memory path of the two-pass kernels is busy with (TA / TCP counters), one-workgroup instance and duo, 20 000 x 5 000
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06
PMC_TAG=big PMC_ENV=PK_OPT_PUB_DUO=0 bash $GRAFT_REPO_ROOT/scripts/gpu_pmc_ta.sh > $GRAFT_REPO_ROOT/gpurun_out/r06/c_pmc_big.log 2>&1
PMC_TAG=duo bash $GRAFT_REPO_ROOT/scripts/gpu_pmc_ta.sh > $GRAFT_REPO_ROOT/gpurun_out/r06/c_pmc_duo.log 2>&1
tail -12 $GRAFT_REPO_ROOT/gpurun_out/r06/c_pmc_big.log; tail -6 $GRAFT_REPO_ROOT/gpurun_out/r06/c_pmc_duo.log
