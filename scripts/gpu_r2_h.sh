cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_assoc.py tests/test_gpu_random_worlds.py -x -q -m gpu > gpurun_out/pytest_h.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/pytest_h.log
if [ $rc -ne 0 ]; then tail -60 gpurun_out/pytest_h.log; exit 1; fi
for o in 1 0 1; do
PK_OPT_OWNER_STEP=$o timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --steps ${ST:-20} --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('owner=$o ms/step %.3f route %s observe %.3f assoc %.3f frac %.3f flagged %s over %s summary %r' % (d['ms_per_step'], r['route'], d['kernel_ms_per_step']['observe'], d['kernel_ms_per_step']['assoc'], r['frac'], r.get('particles_sent_to_general_kernels_last_step'), r.get('candidate_list_overflows_last_step'), d['summary']))"
done
