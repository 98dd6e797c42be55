# ablation of k_step_pub on one box (diagnostic variants; results of ab* are wrong by construction)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in ${AB_LIBS:-libparakeet_slam.so libpk_ab1.so libpk_ab2.so libpk_ab3.so libpk_ab4.so libparakeet_slam.so}; do
PK_BENCH_LIB=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-probes --steps ${AB_STEPS:-12} --warmup 4 $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v ms/step %.4f observe %.4f route %s' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route']))"
done
