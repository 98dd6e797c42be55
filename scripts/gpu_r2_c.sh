# quick: parity of the ML routes at large maps, then configs[2] timings of the one-pass route
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_assoc.py tests/test_gpu_random_worlds.py tests/test_gpu_fullsize.py tests/test_gpu_config2.py -x -q -m gpu > gpurun_out/pytest_c.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/pytest_c.log
if [ $rc -ne 0 ]; then exit 1; fi
for warm in ${WARMS:-0 1}; do
PK_OPT_REGS_WARM=$warm timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('warm=$warm ms/step %.3f route %s observe %.3f assoc %.3f frac %.3f nodup %s known %.3f' % (d['ms_per_step'], r['route'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], r['frac'], r.get('no_resample_probe',{}).get('ml',{}).get('avg_launch_ms'), r['ekf_stage']['avg_launch_ms']))"
done
