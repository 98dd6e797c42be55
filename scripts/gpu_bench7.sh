cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_assoc.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/b_single.json 2> gpurun_out/b1.err; tail -2 gpurun_out/b1.err
python - <<'PY'
import json,glob
for n in ['gpurun_out/b_single.json']:
    try:
        d=json.load(open(n))
        print(n, 'ms/step', round(d['ms_per_step'],3), 'value %.3g'%d['value'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],3))
    except Exception as e: print(n, 'ERR', e, open(n).read()[:300])
PY
done
