# A/B on one box: k_step_pub, every XCD on a contiguous EIGHTH of the particles (libpk_eighths.so, PK_XCD_EIGHTHS) against the XCD's share of
# every turn (32 consecutive particles of 256; default): configs[2] in the driver's window and over 50 steps
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05; mkdir -p $O
for rep in 1 2 3; do for v in libparakeet_slam.so libpk_eighths.so; do
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --steps 20 --warmup 5 > $O/ab_t_$v.c2.$rep.json 2>/dev/null
done; done
for v in libparakeet_slam.so libpk_eighths.so; do
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes > $O/ab_t_$v.c2long.json 2>/dev/null
done
python3 - <<'PY'
import json
O='gpurun_out/r05'
for v in ('libparakeet_slam.so','libpk_eighths.so'):
    for rep in (1,2,3):
        d=json.load(open('%s/ab_t_%s.c2.%d.json'%(O,v,rep))); r=d['roofline']
        print(v,'c2',rep,'ms/step %.3f kernel %.3f frac %.3f summary %r'%(d['ms_per_step'],r['avg_launch_ms'],r['frac'],d['summary']))
    d=json.load(open('%s/ab_t_%s.c2long.json'%(O,v))); r=d['roofline']
    print(v,'50 steps: ms/step %.3f kernel %.3f frac %.3f'%(d['ms_per_step'],r['avg_launch_ms'],r['frac']))
PY
