# round 5, first measurements of the exchange: the planned volume at 2/4/8 ranks, and real 4-rank rehearsals with both placements
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 500 python scripts/gpu_migration_model.py 80000 2000 24 2,4,8 > gpurun_out/r05/migration_model_80000x2000.json 2> gpurun_out/r05/mm1.err; echo "model 2000 rc=$?"
timeout -k 10 500 python scripts/gpu_migration_model.py 160000 5000 14 2,4,8 > gpurun_out/r05/migration_model_160000x5000.json 2> gpurun_out/r05/mm2.err; echo "model 5000 rc=$?"
for pl in contiguous balanced; do
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --steps 20 --warmup 5 --particles 20000 --landmarks 2000 --no-cpu-baseline --no-probes --placement $pl > gpurun_out/r05/rehearsal_4_ranks_20000x2000_$pl.json 2> gpurun_out/r05/reh4_$pl.err; echo "rehearsal 4 $pl rc=$?"
done
python3 - <<'PY'
import json
for n in ('migration_model_80000x2000', 'migration_model_160000x5000'):
    try:
        d = json.load(open('gpurun_out/r05/%s.json' % n))
        for w, v in d['worlds'].items(): print(n, w, v['fraction_of_particles'], 'GB/rank', {k: round(x / 1e9, 3) for k, x in v['bytes_per_rank_and_step'].items()})
    except Exception as e: print(n, 'unreadable', e)
for pl in ('contiguous', 'balanced'):
    try:
        d = json.load(open('gpurun_out/r05/rehearsal_4_ranks_20000x2000_%s.json' % pl))
        print(pl, d['ms_per_step'], d.get('placement'), d['migrated_particles_per_step'], d['migrated_bytes_per_step'] / 1e9)
    except Exception as e: print(pl, 'unreadable', e)
PY
