# quick look: default ML bench (config 1) + config 3 ML, no CPU baseline
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c1 ms/step %.4f observe %.4f assoc %.4f frac %.3f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], d['roofline']['frac']))"
done
timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --particles 100000 --landmarks 2000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 ms/step %.4f observe %.4f assoc %.4f frac %.3f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], d['roofline']['frac']))"
