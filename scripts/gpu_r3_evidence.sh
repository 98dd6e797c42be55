# round-3 evidence at one commit (PK_GIT_SHA is passed in: the snapshot on the box carries no .git):
#   1. per-phase stamps of k_step_pub   2. rocprofv3 --kernel-trace --stats of ONLY the timed filter (driver's window)
#   3. PMC traffic (FETCH / WRITE passes)   4. SQ counters
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r03
ST_P=51200 ST_L=2000 timeout -k 10 300 python scripts/gpu_stamps.py > gpurun_out/r03/stamps_k_step_pub_51200x2000.txt 2>&1; echo "stamps rc=$?"
cd /tmp; rm -rf $R/gpurun_out/prof_ev
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ev -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-probes --steps 20 --warmup 5 > $R/gpurun_out/r03/kernel_trace_bench.json 2> $R/gpurun_out/prof_ev.log; echo "trace rc=$?"
cd $R
f=$(find gpurun_out/prof_ev -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && (echo "# git $PK_GIT_SHA: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary --no-probes --steps 20 --warmup 5 (only the timed filter runs: 25 launches of the step's kernels)"; cat "$f") > gpurun_out/r03/kernel_stats_default_bench.csv
rm -rf gpurun_out/prof_ev
bash scripts/gpu_pmc_traffic2.sh 2>&1 | tail -2; cp gpurun_out/pmc_traffic_100000x2000.json gpurun_out/r03/
bash scripts/gpu_pmc_sq.sh > /dev/null 2>&1; python3 - <<'PY'
import json, os
d = json.load(open('gpurun_out/pmc_sq.json'))
json.dump({"git": os.environ.get("PK_GIT_SHA", "unknown"), "config": "bench.py --steps 3 --warmup 1 --particles 51200 --landmarks 2000 (200 particles per CU)", "counters": d},
          open('gpurun_out/r03/pmc_sq_51200x2000.json', 'w'), indent=1)
k = [x for x in d if 'k_step_pub' in x][0]; c = d[k]
print('k_step_pub VALU per wave.particle %.0f, SALU %.0f, wait %.2f, issue-stall %.2f, active %.2f' % (c['SQ_INSTS_VALU'] / 2048 / 200, c['SQ_INSTS_SALU'] / 2048 / 200, c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']))
PY
head -5 gpurun_out/r03/kernel_stats_default_bench.csv | cut -c1-160; tail -12 gpurun_out/r03/stamps_k_step_pub_51200x2000.txt
