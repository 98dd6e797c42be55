# A/B of library builds on ONE box at the default bench workload (configs[2]): AB_LIBS="libpk_a.so libparakeet_slam.so" [AB_ARGS=...]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for v in $AB_LIBS; do
PK_BENCH_LIB=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --steps ${AB_STEPS:-20} --warmup 5 $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v ms/step %.4f observe %.4f route %s summary %r' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route'], d['summary']))"
done
done
