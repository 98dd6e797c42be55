# A/B on one box: the far lists' variance margin kFarVarFactor 2 (default) / 1.5 / 1.25 where the early steps of big maps are: 20 000 x 5 000
# and a configs[4] shard in the driver's window; configs[2] beside them
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for v in libparakeet_slam.so libpk_vf125.so libpk_vf11.so libpk_vf10.so; do
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 20000 --landmarks 5000 --steps 20 --warmup 5 > $O/ab_s_$v.big.$rep.json 2>/dev/null
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --steps 20 --warmup 5 > $O/ab_s_$v.c2.$rep.json 2>/dev/null
done; done
for v in libparakeet_slam.so libpk_vf125.so libpk_vf11.so libpk_vf10.so; do
PK_BENCH_LIB=$v timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-refscene --no-probes --steps 5 --warmup 2 > $O/ab_s_$v.c4.json 2>/dev/null
done
python3 - <<'PY'
import json
O='gpurun_out/r05'
for v in ("libparakeet_slam.so","libpk_vf125.so","libpk_vf11.so","libpk_vf10.so"):
    for w in ('big','c2'):
        for rep in (1,2):
            d=json.load(open('%s/ab_s_%s.%s.%d.json'%(O,v,w,rep))); r=d['roofline']
            print(v,w,rep,'ms/step %.3f kernel %.3f frac %.3f fallback %s'%(d['ms_per_step'],r['avg_launch_ms'],r['frac'],d.get('particles_sent_to_fallback_kernels_last_step')))
    c=json.load(open('%s/ab_s_%s.c4.json'%(O,v)))['configs4_shard']
    print(v,'configs4 shard: steps 5-24 %.2f ms (%.3f), steps 40-49 %.2f ms (%.3f) fallback %s'%(c['ms_per_step'],c['roofline']['frac'],c['late_window']['ms_per_step'],c['late_window']['roofline']['frac'],c.get('particles_sent_to_fallback_kernels_last_step')))
PY
