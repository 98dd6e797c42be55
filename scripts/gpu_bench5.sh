cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --particles 100000 --landmarks 2000 > gpurun_out/bench_c3_ml.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --particles 20000 --landmarks 1000 > gpurun_out/bench_1k_ml.json 2>/dev/null
python - <<'PY'
import json
for n in ['bench_c3_ml','bench_1k_ml']:
    d=json.load(open(f'gpurun_out/{n}.json')); print(n, round(d['ms_per_step'],3), '%.3g'%d['value'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, round(d['roofline']['frac'],3))
PY
