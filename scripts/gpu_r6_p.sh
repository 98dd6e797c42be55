# where the publish / subscribe instance for maps of at most 512 landmarks (option pub_small) beats k_step_fused: whole-step time over P and L
# (SWEEP="P L;P L;..." overrides the list)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
SWEEP=${SWEEP:-"5000 500;10000 500;20000 500;40000 500;100000 500;10000 256;40000 256;100000 256;10000 128;100000 128"}
IFS=';' read -ra LIST <<< "$SWEEP"
for PL in "${LIST[@]}"; do
set -- $PL
for v in 0 1 0 1 0 1; do PK_OPT_PUB_SMALL=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 120 --warmup 10 --particles $1 --landmarks $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('P $1 L $2 pub_small $v ms/step %.4f observe %.4f route %s' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route']))"; done
done 2>&1 | tee $O/p_pub_small_sweep${SWEEP_TAG}.log
