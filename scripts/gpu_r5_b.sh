cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for pl in contiguous balanced; do
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --steps 20 --warmup 5 --particles 20000 --landmarks 2000 --no-cpu-baseline --no-probes --placement $pl > gpurun_out/r05/rehearsal_4_ranks_20000x2000_$pl.json 2> gpurun_out/r05/reh4_$pl.err; echo "rehearsal 4 $pl rc=$?"
done
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 6 --steps 12 --warmup 3 --particles 20000 --landmarks 5000 --no-cpu-baseline --no-probes > gpurun_out/r05/rehearsal_6_ranks_20000x5000_balanced.json 2> gpurun_out/r05/reh6.err; echo "rehearsal 6 rc=$?"
python3 - <<'PY'
import json
for n in ('rehearsal_4_ranks_20000x2000_contiguous', 'rehearsal_4_ranks_20000x2000_balanced', 'rehearsal_6_ranks_20000x5000_balanced'):
    try:
        d = json.load(open('gpurun_out/r05/%s.json' % n))
        print(n, d['ms_per_step'], d.get('placement'), d['migrated_particles_per_step'], d['migrated_bytes_per_step'] / 1e9, d['roofline'].get('kernel', '')[:30])
    except Exception as e: print(n, 'unreadable', e)
PY
