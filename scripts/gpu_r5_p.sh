# A/B on one box, second look: k_step_pub with the workgroups of one XCD on consecutive particles (default) against the plain deal
# (libpk_noxcd.so, PK_DIAG_NO_XCD_RUNS): configs[2] in the driver's window and over 50 steps, with the probes; configs[1] (pub_small)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05; mkdir -p $O
for rep in 1 2 3; do for v in libpk_noxcd.so libparakeet_slam.so; do
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --steps 20 --warmup 5 > $O/ab_p_$v.c2.$rep.json 2>/dev/null
done; done
for v in libpk_noxcd.so libparakeet_slam.so; do
PK_BENCH_LIB=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene > $O/ab_p_$v.c2long.json 2>/dev/null
PK_BENCH_LIB=$v PK_OPT_PUB_SMALL=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 10000 --landmarks 500 --steps 120 --warmup 10 > $O/ab_p_$v.c1.json 2>/dev/null
done
python3 - <<'PY'
import json
O='gpurun_out/r05'
for v in ('libpk_noxcd.so','libparakeet_slam.so'):
    for rep in (1,2,3):
        d=json.load(open('%s/ab_p_%s.c2.%d.json'%(O,v,rep))); r=d['roofline']
        print(v,'c2',rep,'ms/step %.3f kernel %.3f frac %.3f'%(d['ms_per_step'],r['avg_launch_ms'],r['frac']))
    d=json.load(open('%s/ab_p_%s.c2long.json'%(O,v))); r=d['roofline']
    pr=d.get('no_resample_probe') or {}
    print(v,'50 steps: ms/step %.3f kernel %.3f frac %.3f no-dup %s steady %s'%(d['ms_per_step'],r['avg_launch_ms'],r['frac'],r.get('frac_no_duplicates'),(pr.get('ml_steady_state_map') or {}).get('frac')))
    d=json.load(open('%s/ab_p_%s.c1.json'%(O,v))); r=d['roofline']
    print(v,'configs[1] pub_small: ms/step %.4f kernel %.4f frac %.3f route %s'%(d['ms_per_step'],r['avg_launch_ms'],r['frac'],r['route']))
PY
