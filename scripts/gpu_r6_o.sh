# growing maps at scale, step by step: the one-pass route and the general one (refscene.new_landmarks_at_scale's scene at 20 000 particles)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python scripts/gpu_diag_grow_scale.py 20000 2000 6 16 14 > $O/o_grow_scale.log 2>&1; echo rc $?
GROW_GENERAL=1 timeout -k 10 300 python scripts/gpu_diag_grow_scale.py 20000 2000 6 16 8 > $O/o_grow_scale_general.log 2>&1; echo rc $?
tail -16 $O/o_grow_scale.log; tail -9 $O/o_grow_scale_general.log
