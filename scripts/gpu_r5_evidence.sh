# round-5 evidence at ONE commit.  Run through scripts/run_evidence.sh, which refuses a dirty tree and passes the commit in
# (PK_GIT_SHA: the snapshot on the box carries no .git).  Parts (EV_PARTS, default all):
#   a  stamps per phase and per wave, 15 steps before the measured three (the scan-level pruning needs a map seen a few times):
#      k_step_pub 51 200 x 2 000, k_step_fused 10 000 x 500, k_step_pub_big 20 480 x 5 000 (also 45 steps in: the steady state)
#   b  rocprofv3 --kernel-trace --stats of ONLY the timed filter (the driver's window), configs[2] and 20 000 x 5 000
#   c  PMC traffic (FETCH / WRITE passes) at 100 000 x 2 000, 20 000 x 5 000 and at the size the driver times configs[4]'s shard, 125 000 x 5 000:
#      three early steps of a fresh filter (gpu_pmc_traffic2.sh) and the driver's window (gpu_pmc_window.sh)
#   d  SQ counters over the driver's window (warm-up 5, 20 steps) at 51 200 x 2 000, 20 480 x 5 000, 10 240 x 500
#   e  bench lines: the driver's own command, the default 50 steps, 20 000 x 5 000
#   f  N > 1 rehearsals on the one GPU (gloo): 4 ranks x 20 000 x 2 000 with both placements, 5 ranks x 20 000 x 5 000
#      (a box allows six processes on its card); the planned volume at 2 / 4 / 8 ranks (scripts/gpu_migration_model.py)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05ev; mkdir -p $O
if [ -z "$PK_GIT_SHA" ]; then echo "PK_GIT_SHA is not set: run scripts/run_evidence.sh"; exit 2; fi
echo "$PK_GIT_SHA" > $O/GIT_SHA
PARTS=${EV_PARTS:-abcdef}
if [[ $PARTS == *a* ]]; then
ST_WARM=15 ST_P=51200 ST_L=2000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_51200x2000.txt 2>&1; echo "stamps pub rc=$?"
ST_WARM=15 ST_P=10000 ST_L=500 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_fused_10000x500.txt 2>&1; echo "stamps fused rc=$?"
ST_WARM=15 ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_big_20480x5000.txt 2>&1; echo "stamps big rc=$?"
ST_WARM=45 ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_big_20480x5000_45_steps_in.txt 2>&1; echo "stamps big late rc=$?"
fi
if [[ $PARTS == *b* ]]; then
for cfg in "default:" "20000x5000:--particles 20000 --landmarks 5000"; do
tag=${cfg%%:*}; extra=${cfg#*:}
cd /tmp; rm -rf $R/gpurun_out/prof_ev
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ev -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 20 --warmup 5 $extra > $R/$O/kernel_trace_bench_$tag.json 2> $R/gpurun_out/prof_ev.log; echo "trace $tag rc=$?"
cd $R
f=$(find gpurun_out/prof_ev -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && (echo "# git $PK_GIT_SHA: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 20 --warmup 5 $extra (only the timed filter runs: 25 launches of the step's kernels, the five warm-up steps included)"; cat "$f") > $O/kernel_stats_bench_$tag.csv
rm -rf gpurun_out/prof_ev
done
fi
if [[ $PARTS == *c* ]]; then
bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -2; cp gpurun_out/pmc_traffic_100000x2000.json $O/
PMC_P=20000 PMC_L=5000 bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_20000x5000.json $O/
PMC_P=125000 PMC_L=5000 bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_125000x5000.json $O/
# ... and over the driver's window (bench.py's own 25 launches)
bash scripts/gpu_pmc_window.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_window_100000x2000.json $O/
PMC_P=20000 PMC_L=5000 bash scripts/gpu_pmc_window.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_window_20000x5000.json $O/
PMC_P=125000 PMC_L=5000 bash scripts/gpu_pmc_window.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_window_125000x5000.json $O/
fi
if [[ $PARTS == *d* ]]; then
for cfg in "51200 2000 200" "20480 5000 80" "10240 500 40"; do
set -- $cfg
SQ_STEPS=20 SQ_WARMUP=5 PMC_P=$1 PMC_L=$2 bash scripts/gpu_pmc_sq.sh > /dev/null 2>&1; PP=$1 LL=$2 PER=$3 OO=$O python3 - <<'PY'
import json, os
P, L, per, O = os.environ["PP"], os.environ["LL"], float(os.environ["PER"]), os.environ["OO"]
d = json.load(open('gpurun_out/pmc_sq.json'))
json.dump({"git": os.environ.get("PK_GIT_SHA", "unknown"), "config": "bench.py --steps 20 --warmup 5 --particles %s --landmarks %s (%d particles per CU; counters are means per launch over the 25 launches)" % (P, L, per), "counters": d},
          open('%s/pmc_sq_%sx%s.json' % (O, P, L), 'w'), indent=1)
for k, c in d.items():
    if ('k_step_pub' in k or 'k_step_fused' in k) and c.get('SQ_INSTS_VALU', 0) > 1e6:
        print(k[:40], 'VALU per wave.particle %.0f, SALU %.0f, wait %.2f, issue-stall %.2f, active %.2f' % (c['SQ_INSTS_VALU'] / 2048 / per, c['SQ_INSTS_SALU'] / 2048 / per, c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']))
PY
done
fi
if [[ $PARTS == *e* ]]; then
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_window.json 2> $O/bench_driver_err.txt; echo "bench driver-window rc=$?"
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default_err.txt; echo "bench default rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --steps 20 --warmup 5 --particles 20000 --landmarks 5000 > $O/bench_20000x5000_ml.json 2> $O/bench_20000x5000_err.txt; echo "bench 20000x5000 rc=$?"
fi
if [[ $PARTS == *f* ]]; then
timeout -k 10 500 python scripts/gpu_migration_model.py 80000 2000 24 2,4,8 > $O/migration_model_80000x2000.json 2> $O/mm1.err; echo "model 2000 rc=$?"
timeout -k 10 500 python scripts/gpu_migration_model.py 160000 5000 24 2,4,8 > $O/migration_model_160000x5000.json 2> $O/mm2.err; echo "model 5000 rc=$?"
for pl in contiguous balanced; do
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --steps 20 --warmup 5 --particles 20000 --landmarks 2000 --no-cpu-baseline --no-probes --placement $pl > $O/rehearsal_4_ranks_20000x2000_$pl.json 2> $O/reh4_$pl.err; echo "rehearsal 4 $pl rc=$?"
done
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 5 --steps 12 --warmup 5 --particles 20000 --landmarks 5000 --no-cpu-baseline --no-probes > $O/rehearsal_5_ranks_20000x5000_balanced.json 2> $O/reh5.err; echo "rehearsal 5 rc=$?"
fi
python3 - <<'PY'
import json, glob, os
O = 'gpurun_out/r05ev'
for n in ('bench_driver_window', 'bench_default', 'bench_20000x5000_ml'):
    try:
        d = json.load(open('%s/%s.json' % (O, n))); r = d['roofline']
        print(n, 'ms/step %.3f value %.4g kernel %.3f ms frac %.3f bound %s issue %s no-dup %s traffic %s' % (d['ms_per_step'], d['value'], r['avg_launch_ms'], r['frac'], r['bound'], (r.get('issue') or {}).get('frac'), r.get('frac_no_duplicates'), r.get('traffic')))
        for k in ('configs1', 'configs4_shard'):
            if d.get(k): print('  ', k, d[k].get('ms_per_step'), (d[k].get('roofline') or {}).get('frac'), (d[k].get('late_window') or {}).get('ms_per_step'), d[k].get('error'))
    except Exception as e:
        print(n, 'unreadable', e)
for f in sorted(glob.glob(O + '/rehearsal_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d.get('placement'), d['migrated_particles_per_step'], d['migrated_bytes_per_step'] / 1e9)
    except Exception as e:
        print(f, 'unreadable', e)
PY
