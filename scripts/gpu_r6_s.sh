# bench.py --gpus N rehearsed on ONE GPU (gloo, ranks share the device): the N > 1 line's fields after round 6's changes to the bench;
# N = 2 at the map size of configs[2] and of configs[4] (the two-pass route), N = 4 at configs[2]'s; and --gpus 1 --force-sharded over nccl
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
for cfg in "2 20000 2000" "4 20000 2000" "2 8000 5000"; do
set -- $cfg
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus $1 --steps 20 --warmup 5 --particles $2 --landmarks $3 --no-cpu-baseline > $O/s_bench_n$1_$2x$3.json 2> $O/s_bench_n$1_$2x$3.err; echo "rc $?"; tail -2 $O/s_bench_n$1_$2x$3.err
python - <<PY
import json
d=json.load(open('$O/s_bench_n$1_$2x$3.json'))
print('n', $1, 'world', d.get('world_size'), 'rehearsal', d.get('rehearsal'), 'ms/step %.3f' % d['ms_per_step'], 'placement', d.get('placement'), 'migrated/step', d.get('migrated_particles_per_step'), 'of', d['config']['global_particles'], 'route', d['roofline']['route'], 'warmup', d.get('warmup_steps', {}).get('ms'))
PY
done
timeout -k 10 500 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --no-cpu-baseline > $O/s_bench_n1_sharded.json 2> $O/s_bench_n1_sharded.err; echo "rc $?"; tail -2 $O/s_bench_n1_sharded.err
python - <<PY
import json
d=json.load(open('$O/s_bench_n1_sharded.json'))
print('n 1 sharded', d['config']['parallelism'][:80], 'ms/step %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])
PY
