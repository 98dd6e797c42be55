cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for nv in 1 2; do
PK_OBSERVE_NV=$nv timeout 600 python bench.py --no-cpu-baseline --assoc known --steps 6 --warmup 2 --particles 100000 --landmarks 2000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('known nv $nv c3 ms/step %.4f observe %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe']))"
done
