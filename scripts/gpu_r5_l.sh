cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_pub.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_assoc.py tests/test_gpu_audit.py -x -q -m gpu > gpurun_out/r05/t_l.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05/t_l.log
cd /tmp
for ps in 1; do
rm -rf $R/gpurun_out/prof_k
PK_OPT_PUB_SMALL=$ps timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_k -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 10000 --landmarks 500 --steps 120 --warmup 10 > /dev/null 2> $R/gpurun_out/prof_k.log
f=$(find $R/gpurun_out/prof_k -name '*kernel_stats.csv' | head -1); cp $f $R/gpurun_out/r05/kstats_c1_ps${ps}_b.csv
cut -c1-110 $f | head -8
done
rm -rf $R/gpurun_out/prof_k
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_k -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --steps 20 --warmup 5 > /dev/null 2> $R/gpurun_out/prof_k.log
f=$(find $R/gpurun_out/prof_k -name '*kernel_stats.csv' | head -1); cut -c1-110 $f | head -8
rm -rf $R/gpurun_out/prof_k
cd $R
for rep in 1 2; do for ps in 0 1; do
PK_OPT_PUB_SMALL=$ps timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 10000 --landmarks 500 --steps 120 --warmup 10 > gpurun_out/r05/c1_ps$ps.$rep.json 2>/dev/null
done; done
python3 - <<'PY'
import json
for rep in (1,2):
  for ps in (0,1):
    d = json.load(open('gpurun_out/r05/c1_ps%d.%d.json' % (ps, rep))); r = d['roofline']
    print('pub_small', ps, 'ms/step %.4f kernel %.4f frac %.3f route %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['route']))
PY
