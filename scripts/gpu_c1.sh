cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c1 ms/step %.4f observe %.4f assoc %.4f frac %.3f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], d['roofline']['frac']))"
done
