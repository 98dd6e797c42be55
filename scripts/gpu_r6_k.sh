# round 6: growing maps on the one-pass routes -- the grow / new-landmark tests (and the multi-rank ones that grow), then configs[2] and configs[1]
# with the library before the change beside the one with it (three interleaved repetitions)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_grow.py tests/test_gpu_new_landmarks.py tests/test_gpu_facade.py tests/test_gpu_pub.py -q -m gpu > $O/k_tests.log 2>&1; rc=$?; echo "tests rc $rc" | tee -a $O/k_tests.log
tail -25 $O/k_tests.log
if grep -q "Memory access fault" $O/k_tests.log; then echo "FAULT"; exit 1; fi
AB_LIBS="libpk_prev.so libparakeet_slam.so" AB_TAG=unm_rows bash scripts/gpu_ab3.sh 2>&1 | tail -8
cp gpurun_out/r04/ab3_unm_rows.log $O/k_ab3_configs2.log
AB_LIBS="libpk_prev.so libparakeet_slam.so" AB_TAG=unm_rows_c1 AB_ARGS="--particles 10000 --landmarks 500" AB_STEPS=120 bash scripts/gpu_ab3.sh 2>&1 | tail -8
cp gpurun_out/r04/ab3_unm_rows_c1.log $O/k_ab3_configs1.log
for v in 0 1; do PK_OPT_PUB_SMALL=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 120 --warmup 10 --particles 10000 --landmarks 500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pub_small $v ms/step %.4f observe %.4f route %s' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route']))"; done
for v in 0 1; do PK_OPT_PUB_SMALL=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 120 --warmup 10 --particles 10000 --landmarks 500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pub_small $v ms/step %.4f observe %.4f route %s' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route']))"; done
