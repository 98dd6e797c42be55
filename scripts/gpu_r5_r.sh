# A/B on one box: k_step_pub_big with runs of consecutive particles per XCD (tuning builds: libpk_run4.so, libpk_run8.so = -DPK_BIG_XCD_RUN=4 / 8; libpk_beighths.so = -DPK_BIG_EIGHTHS) against the plain deal
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for v in libparakeet_slam.so libpk_beighths.so; do
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 20000 --landmarks 5000 --steps 20 --warmup 5 > $O/ab_r_$v.big.$rep.json 2>/dev/null
done; done
for v in libparakeet_slam.so libpk_beighths.so; do
PK_BENCH_LIB=$v timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-refscene --no-probes --steps 5 --warmup 2 > $O/ab_r_$v.c4.json 2>/dev/null
done
python3 - <<'PY'
import json
O='gpurun_out/r05'
for v in ("libparakeet_slam.so","libpk_beighths.so"):
    for rep in (1,2):
        d=json.load(open('%s/ab_r_%s.big.%d.json'%(O,v,rep))); r=d['roofline']
        print(v,'20000x5000',rep,'ms/step %.3f kernel %.3f frac %.3f'%(d['ms_per_step'],r['avg_launch_ms'],r['frac']))
    c=json.load(open('%s/ab_r_%s.c4.json'%(O,v)))['configs4_shard']
    print(v,'configs4 shard: steps 5-24 %.2f ms (%.3f), steps 40-49 %.2f ms (%.3f)'%(c['ms_per_step'],c['roofline']['frac'],c['late_window']['ms_per_step'],c['late_window']['roofline']['frac']))
PY
