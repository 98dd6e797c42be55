# round 6: why two and three workgroups per CU lose -- what the L2s pass on to the fabric (FETCH_SIZE / WRITE_SIZE over bench.py's own window) with
# one, two and three particles in flight per CU, 20 000 x 5 000
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
for v in 0 1 2; do
PK_OPT_PUB_DUO=$v PMC_P=20000 PMC_L=5000 bash scripts/gpu_pmc_window.sh 2>&1 | tail -1
cp gpurun_out/pmc_traffic_window_20000x5000.json $O/j_traffic_window_20000x5000_pub_duo_$v.json
done
python3 - <<'PY'
import json
for v in (0,1,2):
    d=json.load(open('gpurun_out/r06/j_traffic_window_20000x5000_pub_duo_%d.json'%v))
    print('pub_duo', v, {k: round(x,3) for k,x in d['x_algorithmic'].items()}, {k: [round(y[0]/1e6,2), y[1]] for k,y in d['raw_kib_mean_and_launches'].items() for y in [y.get('FETCH_SIZE',[0,0])]})
PY
