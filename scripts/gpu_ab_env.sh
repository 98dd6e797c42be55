# A/B of one tuning environment variable on ONE box: AB_VAR=PK_PREFETCH AB_VALUES="0 512 1024" [AB_ARGS="--particles ..."]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for v in $AB_VALUES; do
env $AB_VAR=$v timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$AB_VAR=$v ms/step %.4f observe %.4f assoc %.4f frac %.3f known %.4f summary %r matched %r' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], d['roofline']['frac'], d.get('ekf_stage_supplied_ids', {}).get('avg_launch_ms', 0), d['summary'], d['matched_fraction_last_timed_scan']))"
done
done
