# configs[1]: k_step_fused against the 256-lane publish / subscribe instance (option pub_small), same box
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 0 1 0 1; do
PK_OPT_PUB_SMALL=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-probes --steps ${AB_STEPS:-200} --warmup 20 --particles 10000 --landmarks 500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pub_small=$v ms/step %.4f observe %.4f route %s frac %.3f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route'], d['roofline']['frac']))"
done
