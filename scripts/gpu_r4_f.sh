# round 4: configs[1] (10 000 x 500) -- k_step_fused against the 256-lane publish / subscribe instance ("pub_small"), three repetitions
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for rep in 1 2 3; do
for v in 0 1; do
PK_OPT_PUB_SMALL=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --particles 10000 --landmarks 500 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pub_small=$v ms/step %.4f observe %.4f assoc %.4f route %s frac %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], d['roofline']['route'], d['roofline']['frac']))"
done
done 2>&1 | tee gpurun_out/r04/ab_configs1_pub_small.log
