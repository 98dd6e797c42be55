# round-3 evidence, third part: the L > 2 048 route under rocprofv3 (kernel trace of only the timed filter; SQ counters)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r03
cd /tmp; rm -rf $R/gpurun_out/prof_ev3
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ev3 -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-probes --steps 20 --warmup 5 --particles 20000 --landmarks 5000 > $R/gpurun_out/r03/kernel_trace_bench_20000x5000.json 2> $R/gpurun_out/prof_ev3.log; echo "trace rc=$?"
cd $R
f=$(find gpurun_out/prof_ev3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && (echo "# git $PK_GIT_SHA: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary --no-probes --steps 20 --warmup 5 --particles 20000 --landmarks 5000"; cat "$f") > gpurun_out/r03/kernel_stats_20000x5000.csv
rm -rf gpurun_out/prof_ev3
PMC_P=20480 PMC_L=5000 bash scripts/gpu_pmc_sq.sh > /dev/null 2>&1; python3 - <<'PY'
import json, os
d = json.load(open('gpurun_out/pmc_sq.json'))
json.dump({"git": os.environ.get("PK_GIT_SHA", "unknown"), "config": "bench.py --steps 3 --warmup 1 --particles 20480 --landmarks 5000 (80 particles per CU)", "counters": d},
          open('gpurun_out/r03/pmc_sq_20480x5000.json', 'w'), indent=1)
k = [x for x in d if 'k_step_pub_big' in x][0]; c = d[k]
print('k_step_pub_big VALU per wave.particle %.0f, SALU %.0f, wait %.2f, issue-stall %.2f, active %.2f, VALU busy %.2f' % (c['SQ_INSTS_VALU'] / 2048 / 80, c['SQ_INSTS_SALU'] / 2048 / 80, c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_VALU'] * 4 / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)))
PY
head -4 gpurun_out/r03/kernel_stats_20000x5000.csv | cut -c1-150
python3 -c "
import json; d=json.load(open('gpurun_out/r03/kernel_trace_bench_20000x5000.json')); print('bench under rocprof: ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
