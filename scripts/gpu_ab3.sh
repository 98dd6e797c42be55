# A/B of library builds on ONE box, three interleaved repetitions: AB_LIBS="a.so b.so ..." [AB_ARGS=...]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for rep in 1 2 3; do
for v in $AB_LIBS; do
PK_BENCH_LIB=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps ${AB_STEPS:-20} --warmup 5 $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v ms/step %.4f observe %.4f route %s summary %.12g' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route'], d['summary'][0]))"
done
done 2>&1 | tee gpurun_out/r04/ab3_${AB_TAG:-x}.log
