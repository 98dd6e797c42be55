# round 6, first call: what the publish table comes to along the bench's trajectory at 5 000 and 2 000 landmarks (every step, step 0 included)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
ST_P=20000 ST_L=5000 ST_S=50 ST_OUT=$O/pubstats_20000x5000.json timeout -k 10 400 python scripts/gpu_diag_pubstats.py > $O/pubstats_20000x5000.log 2>&1 &&
ST_P=20000 ST_L=2000 ST_S=30 ST_OUT=$O/pubstats_20000x2000.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/pubstats_20000x2000.log 2>&1 &&
ST_P=10000 ST_L=500 ST_S=30 ST_OUT=$O/pubstats_10000x500.json timeout -k 10 300 python scripts/gpu_diag_pubstats.py > $O/pubstats_10000x500.log 2>&1
tail -3 $O/pubstats_20000x5000.log
