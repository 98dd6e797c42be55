# Rehearsal of bench.py's N > 1 control flow on ONE GPU: two ranks share cuda:0, gloo instead of RCCL
# (the collectives stage through the host: the timing means nothing, the flow and the result do).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PK_BENCH_SAME_GPU=1 PK_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 3 2>gpurun_out/bench2.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks (gloo, one GPU): ms/step %.3f value %.3g n_gpus %d summary %r parallelism %s' % (d['ms_per_step'], d['value'], d['n_gpus'], d['summary'], d['config']['parallelism'][:60]))"
tail -3 gpurun_out/bench2.err
