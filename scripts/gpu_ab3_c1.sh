# A/B of library builds at configs[1] (10 000 x 500), three interleaved repetitions: AB_LIBS="a.so b.so ..."
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for rep in 1 2 3; do
for v in $AB_LIBS; do
PK_BENCH_LIB=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --particles 10000 --landmarks 500 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v ms/step %.4f observe %.4f route %s frac %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route'], d['roofline']['frac']))"
done
done 2>&1 | tee gpurun_out/r04/ab3_c1_${AB_TAG:-x}.log
