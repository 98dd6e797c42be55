cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/b_single.json 2> gpurun_out/b.err; tail -2 gpurun_out/b.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --force-sharded > gpurun_out/b_sharded1.json 2> gpurun_out/b.err; tail -2 gpurun_out/b.err
python - <<'PY'
import json,glob
for n in ['gpurun_out/b_single.json','gpurun_out/b_sharded1.json']:
    try:
        d=json.load(open(n))
        print(n, 'ms/step', round(d['ms_per_step'],3), 'value %.3g'%d['value'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],3))
    except Exception as e: print(n, 'ERR', e, open(n).read()[:300])
PY
