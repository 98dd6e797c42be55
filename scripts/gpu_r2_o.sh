# how many particles cross rank boundaries per resample?  two and four ranks on ONE GPU over gloo (rehearsal switches)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 2 4; do
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout 800 python bench.py --gpus $n --steps 20 --warmup 5 --particles 20000 --landmarks 2000 --no-cpu-baseline > gpurun_out/o_bench$n.json 2> gpurun_out/o_bench$n.err; echo "rc $?"; tail -2 gpurun_out/o_bench$n.err
python - <<PY
import json
d=json.load(open('gpurun_out/o_bench$n.json'))
print('n_gpus', d['n_gpus'], 'ms/step', d['ms_per_step'], 'migrated/step', d.get('migrated_particles_per_step'), 'of', d['config']['global_particles'], 'bytes/step %.3g' % d.get('migrated_bytes_per_step', 0))
PY
done
