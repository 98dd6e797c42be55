cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for nv in 1 2; do
PK_OBSERVE_NV=$nv timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ml_nv$nv.json 2> gpurun_out/bench_ml.err; tail -2 gpurun_out/bench_ml.err
PK_OBSERVE_NV=$nv timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --assoc known > gpurun_out/bench_known_nv$nv.json 2> gpurun_out/bench_ml.err; tail -2 gpurun_out/bench_ml.err
PK_OBSERVE_NV=$nv timeout 600 python bench.py --steps 10 --warmup 3 --particles 100000 --landmarks 2000 --no-cpu-baseline > gpurun_out/bench_c3_ml_nv$nv.json 2> gpurun_out/bench_c3_ml.err; tail -2 gpurun_out/bench_c3_ml.err
done
python - <<'PY'
import json,glob
for n in sorted(glob.glob('gpurun_out/bench_*nv*.json')):
    try:
        d=json.load(open(n))
        print(n, 'ms/step', round(d['ms_per_step'],3), 'value %.3g'%d['value'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],3))
    except Exception as e: print(n, 'ERR', e)
PY
