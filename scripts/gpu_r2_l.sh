# regs_warm A/B on the no-spill k_step_regs: AB_LIBS x warm
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $AB_LIBS; do for w in ${AB_WARM:-0 1}; do
PK_BENCH_LIB=$v PK_OPT_REGS_WARM=$w timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v warm $w ms/step %.4f observe %.4f' % (d['ms_per_step'], d['kernel_ms_per_step']['observe']))"
done; done; done
