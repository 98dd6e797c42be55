# A/B on one box: the two-pass kernel's pass 2 back to front (libparakeet_slam.so) against front to back (libpk_f2b.so):
# 20 000 x 5 000 in the driver's window and 45 steps in; 125 000 x 5 000 steps 5-24 and 40-49 (configs4_shard)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for v in libpk_f2b.so libparakeet_slam.so; do
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 20000 --landmarks 5000 --steps 20 --warmup 5 > $O/ab_n_$v.early.$rep.json 2>/dev/null
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 20000 --landmarks 5000 --steps 10 --warmup 45 > $O/ab_n_$v.late.$rep.json 2>/dev/null
done; done
for v in libpk_f2b.so libparakeet_slam.so; do
PK_BENCH_LIB=$v timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --no-refscene --no-probes --steps 5 --warmup 2 > $O/ab_n_$v.c4.json 2>/dev/null
done
python3 - <<'PY'
import json
O='gpurun_out/r05'
for v in ('libpk_f2b.so','libparakeet_slam.so'):
    for w in ('early','late'):
        for rep in (1,2):
            d=json.load(open('%s/ab_n_%s.%s.%d.json'%(O,v,w,rep))); r=d['roofline']
            print(v,w,rep,'ms/step %.3f kernel %.3f frac %.3f'%(d['ms_per_step'],r['avg_launch_ms'],r['frac']))
    c=json.load(open('%s/ab_n_%s.c4.json'%(O,v)))['configs4_shard']
    print(v,'configs4 shard: steps 5-24 %.2f ms (%.3f), steps 40-49 %.2f ms (%.3f)'%(c['ms_per_step'],c['roofline']['frac'],c['late_window']['ms_per_step'],c['late_window']['roofline']['frac']))
PY
