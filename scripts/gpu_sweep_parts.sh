cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_assoc.py -x -q -m gpu 2>&1 | tail -3
for dbg in 0 1 2 3; do
PK_SWEEP_DEBUG=$dbg timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --particles 100000 --landmarks 2000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg $dbg c3 ms/step %.4f observe %.4f assoc %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc']))"
done
