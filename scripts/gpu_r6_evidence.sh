# round-6 evidence at ONE commit.  Run through scripts/run_evidence.sh, which refuses a dirty tree and passes the commit in
# (PK_GIT_SHA: the snapshot on the box carries no .git).  Every counter file gets `code`: the hashes of the kernels' instructions in the
# library that ran (parakeet_slam_amd/codeobj.py) -- bench.py replays a counter only on exactly those instructions.  Parts (EV_PARTS):
#   a  stamps per wave, 15 steps before the measured three: k_step_pub 51 200 x 2 000, k_step_pub_big 20 480 x 5 000 (also 45 steps in), and the
#      same with two (pub_duo = 1) and three (pub_duo = 2) workgroups per CU
#   b  rocprofv3 --kernel-trace --stats of ONLY the timed filter (the driver's window), configs[2] and 20 000 x 5 000
#   c  PMC traffic (FETCH / WRITE passes): three early steps of a fresh filter and the driver's window, 100 000 x 2 000 and 20 000 x 5 000
#   k  ... and at configs[1], 10 000 x 500
#   C  ... and at the size the driver times configs[4]'s shard, 125 000 x 5 000 (a call of its own: 170 GB to fill)
#   d  SQ counters over the driver's window at 51 200 x 2 000, 20 480 x 5 000, 10 240 x 500
#   t  vector-memory path counters (TA / TCP) of the two-pass kernels at 20 000 x 5 000, one / two / three workgroups per CU; fabric traffic of the same three
#   e  bench lines: the driver's own command, 20 000 x 5 000
#   E  the default 50 steps
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06ev; mkdir -p $O
if [ -z "$PK_GIT_SHA" ]; then echo "PK_GIT_SHA is not set: run scripts/run_evidence.sh"; exit 2; fi
echo "$PK_GIT_SHA" > $O/GIT_SHA
python -m parakeet_slam_amd.codeobj --json > $O/kernel_code_hashes.json
PARTS=${EV_PARTS:-abcdte}
if [[ $PARTS == *a* ]]; then
ST_WARM=15 ST_P=51200 ST_L=2000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_51200x2000.txt 2>&1; echo "stamps pub rc=$?"
ST_WARM=15 ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_big_20480x5000.txt 2>&1; echo "stamps big rc=$?"
ST_WARM=45 ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_big_20480x5000_45_steps_in.txt 2>&1; echo "stamps big late rc=$?"
PK_OPT_PUB_DUO=1 ST_WARM=15 ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_duo_20480x5000_two_workgroups_per_cu.txt 2>&1; echo "stamps duo rc=$?"
PK_OPT_PUB_DUO=2 ST_WARM=15 ST_P=20480 ST_L=5000 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_duo_20480x5000_three_workgroups_per_cu.txt 2>&1; echo "stamps trio rc=$?"
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
bash scripts/gpu_pmc_window.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_window_100000x2000.json $O/
PMC_P=20000 PMC_L=5000 bash scripts/gpu_pmc_window.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_window_20000x5000.json $O/
fi
if [[ $PARTS == *k* ]]; then  # ... and at configs[1] (k_step_fused)
PMC_P=10000 PMC_L=500 bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_10000x500.json $O/
PMC_P=10000 PMC_L=500 bash scripts/gpu_pmc_window.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_window_10000x500.json $O/
fi
if [[ $PARTS == *C* ]]; then
PMC_P=125000 PMC_L=5000 bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_125000x5000.json $O/
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
if [[ $PARTS == *t* ]]; then
for v in 0 1 2; do
PMC_TAG=pub_duo_$v PMC_ENV=PK_OPT_PUB_DUO=$v bash scripts/gpu_pmc_ta.sh > $O/pmc_ta_log_pub_duo_$v.txt 2>&1; cp gpurun_out/r06/pmc_ta_pub_duo_$v.json $O/pmc_ta_20000x5000_pub_duo_$v.json
PK_OPT_PUB_DUO=$v PMC_P=20000 PMC_L=5000 bash scripts/gpu_pmc_window.sh 2>&1 | tail -1; cp gpurun_out/pmc_traffic_window_20000x5000.json $O/ab_traffic_window_20000x5000_pub_duo_$v.json
done
fi
if [[ $PARTS == *e* ]]; then
timeout -k 10 1000 python bench.py --steps 20 --warmup 5 > $O/bench_driver_window.json 2> $O/bench_driver_err.txt; echo "bench driver-window rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --steps 20 --warmup 5 --particles 20000 --landmarks 5000 > $O/bench_20000x5000_ml.json 2> $O/bench_20000x5000_err.txt; echo "bench 20000x5000 rc=$?"
fi
if [[ $PARTS == *E* ]]; then
timeout -k 10 1000 python bench.py > $O/bench_default.json 2> $O/bench_default_err.txt; echo "bench default rc=$?"
fi
# every counter file of this run says which instructions it measured
python3 - <<'PY'
import json, glob, os
O = 'gpurun_out/r06ev'
code = json.load(open(O + '/kernel_code_hashes.json'))
for f in sorted(glob.glob(O + '/pmc_*.json') + glob.glob(O + '/ab_traffic_*.json')):
    try:
        d = json.load(open(f))
    except ValueError:
        continue
    if isinstance(d, dict) and 'code' not in d:
        d['code'] = code
        d.setdefault('git', os.environ.get('PK_GIT_SHA', 'unknown'))
        json.dump(d, open(f, 'w'), indent=1)
for n in ('bench_driver_window', 'bench_default', 'bench_20000x5000_ml'):
    try:
        d = json.load(open('%s/%s.json' % (O, n))); r = d['roofline']
        print(n, 'ms/step %.3f value %.4g kernel %.3f ms frac %.3f bound %s issue %s no-dup %s traffic %s window %s' % (d['ms_per_step'], d['value'], r['avg_launch_ms'], r['frac'], r['bound'], (r.get('issue') or {}).get('frac'), r.get('frac_no_duplicates'), r.get('traffic'), r.get('traffic_window')))
        for k in ('configs1', 'configs4_shard'):
            if d.get(k): print('  ', k, d[k].get('ms_per_step'), (d[k].get('roofline') or {}).get('frac'), (d[k].get('late_window') or {}).get('ms_per_step'), d[k].get('error'))
    except Exception as e:
        print(n, 'unreadable', e)
PY
