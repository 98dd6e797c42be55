# long fuzz runs at the round's last kernels: random scenes through the production routes against the general kernels (maps bit for bit),
# the default instances and the two optional instances of the two-pass kernel (pub_duo = 1, 2)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
rc=0
FUZZ_N=1000 FUZZ_SEED=71 timeout -k 10 900 python scripts/gpu_fuzz_routes.py > $O/fuzz_routes_seed71.log 2>&1 || rc=1; tail -n 1 $O/fuzz_routes_seed71.log
FUZZ_OPTS=pub_duo=1 FUZZ_N=600 FUZZ_SEED=72 timeout -k 10 900 python scripts/gpu_fuzz_routes.py > $O/fuzz_routes_seed72_pub_duo_1.log 2>&1 || rc=1; tail -n 1 $O/fuzz_routes_seed72_pub_duo_1.log
FUZZ_OPTS=pub_duo=2 FUZZ_N=600 FUZZ_SEED=73 timeout -k 10 900 python scripts/gpu_fuzz_routes.py > $O/fuzz_routes_seed73_pub_duo_2.log 2>&1 || rc=1; tail -n 1 $O/fuzz_routes_seed73_pub_duo_2.log
FUZZ_N=300 FUZZ_SEED=74 timeout -k 10 900 python scripts/gpu_fuzz_steps.py > $O/fuzz_steps_seed74.log 2>&1 || rc=1; tail -n 1 $O/fuzz_steps_seed74.log
FUZZ_OPTS=pub_duo=1 FUZZ_N=200 FUZZ_SEED=75 timeout -k 10 900 python scripts/gpu_fuzz_steps.py > $O/fuzz_steps_seed75_pub_duo_1.log 2>&1 || rc=1; tail -n 1 $O/fuzz_steps_seed75_pub_duo_1.log
FUZZ_OPTS=pub_duo=2 FUZZ_N=200 FUZZ_SEED=76 timeout -k 10 900 python scripts/gpu_fuzz_steps.py > $O/fuzz_steps_seed76_pub_duo_2.log 2>&1 || rc=1; tail -n 1 $O/fuzz_steps_seed76_pub_duo_2.log
FUZZ_SMALL=1 FUZZ_OPTS=pub_small=1 FUZZ_N=600 FUZZ_SEED=77 timeout -k 10 900 python scripts/gpu_fuzz_routes.py > $O/fuzz_routes_seed77_small_maps_pub_small_1.log 2>&1 || rc=1; tail -n 1 $O/fuzz_routes_seed77_small_maps_pub_small_1.log
FUZZ_SMALL=1 FUZZ_OPTS=pub_small=1 FUZZ_N=200 FUZZ_SEED=78 timeout -k 10 900 python scripts/gpu_fuzz_steps.py > $O/fuzz_steps_seed78_small_maps_pub_small_1.log 2>&1 || rc=1; tail -n 1 $O/fuzz_steps_seed78_small_maps_pub_small_1.log
if grep -q "Memory access fault" $O/fuzz_*.log; then echo FAULT; rc=1; fi
exit $rc
