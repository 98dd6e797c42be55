# round-4 evidence at one commit (PK_GIT_SHA is passed in: the snapshot on the box carries no .git).  Parts (EV_PARTS, default all):
#   a  stamps: k_step_pub per phase and per wave (51 200 x 2 000), k_step_fused (10 000 x 500), k_step_pub_big per wave (20 480 x 5 000)
#   b  rocprofv3 --kernel-trace --stats of ONLY the timed filter (driver's window), at configs[2] and at 20 000 x 5 000
#   c  PMC traffic (FETCH / WRITE passes) at 100 000 x 2 000 and 20 000 x 5 000
#   d  SQ counters at 51 200 x 2 000 and 20 480 x 5 000
#   e  bench lines: driver window (the driver's own command), default 50 steps, 20 000 x 5 000
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04
PARTS=${EV_PARTS:-abcde}
if [[ $PARTS == *a* ]]; then
ST_P=51200 ST_L=2000 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r04/stamps_k_step_pub_51200x2000.txt 2>&1; echo "stamps pub rc=$?"
ST_P=10000 ST_L=500 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r04/stamps_k_step_fused_10000x500.txt 2>&1; echo "stamps fused rc=$?"
ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r04/stamps_k_step_pub_big_20480x5000.txt 2>&1; echo "stamps big rc=$?"
fi
if [[ $PARTS == *b* ]]; then
for cfg in "default:" "20000x5000:--particles 20000 --landmarks 5000"; do
tag=${cfg%%:*}; extra=${cfg#*:}
cd /tmp; rm -rf $R/gpurun_out/prof_ev
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ev -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 20 --warmup 5 $extra > $R/gpurun_out/r04/kernel_trace_bench_$tag.json 2> $R/gpurun_out/prof_ev.log; echo "trace $tag rc=$?"
cd $R
f=$(find gpurun_out/prof_ev -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && (echo "# git $PK_GIT_SHA: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 20 --warmup 5 $extra (only the timed filter runs: 25 launches of the step's kernels)"; cat "$f") > gpurun_out/r04/kernel_stats_bench_$tag.csv
rm -rf gpurun_out/prof_ev
done
fi
if [[ $PARTS == *c* ]]; then
bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -2; cp gpurun_out/pmc_traffic_100000x2000.json gpurun_out/r04/
PMC_P=20000 PMC_L=5000 bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_20000x5000.json gpurun_out/r04/
fi
if [[ $PARTS == *d* ]]; then
for cfg in "51200 2000 200" "20480 5000 80" "10240 500 40"; do
set -- $cfg
PMC_P=$1 PMC_L=$2 bash scripts/gpu_pmc_sq.sh > /dev/null 2>&1; PP=$1 LL=$2 PER=$3 python3 - <<'PY'
import json, os
P, L, per = os.environ["PP"], os.environ["LL"], float(os.environ["PER"])
d = json.load(open('gpurun_out/pmc_sq.json'))
json.dump({"git": os.environ.get("PK_GIT_SHA", "unknown"), "config": "bench.py --steps 3 --warmup 1 --particles %s --landmarks %s (%d particles per CU)" % (P, L, per), "counters": d},
          open('gpurun_out/r04/pmc_sq_%sx%s.json' % (P, L), 'w'), indent=1)
for k, c in d.items():
    if 'k_step_pub' in k and c.get('SQ_INSTS_VALU', 0) > 1e6:
        print(k[:40], 'VALU per wave.particle %.0f, SALU %.0f, wait %.2f, issue-stall %.2f, active %.2f' % (c['SQ_INSTS_VALU'] / 2048 / per, c['SQ_INSTS_SALU'] / 2048 / per, c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']))
PY
done
fi
if [[ $PARTS == *e* ]]; then
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_window.json 2> gpurun_out/r04/bench_driver_err.txt; echo "bench driver-window rc=$?"
timeout -k 10 900 python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default_err.txt; echo "bench default rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --steps 20 --warmup 5 --particles 20000 --landmarks 5000 > gpurun_out/r04/bench_20000x5000_ml.json 2> gpurun_out/r04/bench_20000x5000_err.txt; echo "bench 20000x5000 rc=$?"
python3 - <<'PY'
import json
for n in ('bench_driver_window', 'bench_default', 'bench_20000x5000_ml'):
    try:
        d = json.load(open('gpurun_out/r04/%s.json' % n)); r = d['roofline']
        print(n, 'ms/step %.3f value %.4g kernel %.3f ms frac %.3f bound %s issue %s no-dup %s traffic %s' % (d['ms_per_step'], d['value'], r['avg_launch_ms'], r['frac'], r['bound'], (r.get('issue') or {}).get('frac'), r.get('frac_no_duplicates'), r.get('traffic')))
        if d.get('per_step'): print('   per_step', json.dumps(d['per_step'])[:700])
        if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
        for k in ('configs1', 'configs4_shard'):
            if d.get(k): print('  ', k, d[k].get('ms_per_step'), (d[k].get('roofline') or {}).get('frac'), d[k].get('error'))
        if d.get('refscene'): print('   refscene', {k: v.get('seconds_per_step') for k, v in d['refscene'].items() if isinstance(v, dict)})
    except Exception as e:
        print(n, 'unreadable', e)
PY
fi
