# config-5-like map size on one GPU: 20 000 particles x 5 000 landmarks (23 GB of maps), ML and supplied ids
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --steps 4 --warmup 2 --particles 20000 --landmarks 5000 > gpurun_out/bench_c5like_ml.json 2> gpurun_out/bench_c5like_ml.err; tail -2 gpurun_out/bench_c5like_ml.err
python -c "import json; d=json.load(open('gpurun_out/bench_c5like_ml.json')); print('c5like ML ms/step %.3f observe %.3f assoc %.3f value %.3g' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], d['value'])); k=d.get('ekf_stage_supplied_ids'); print('supplied ids', k and (k['avg_launch_ms'], k['frac']))"
