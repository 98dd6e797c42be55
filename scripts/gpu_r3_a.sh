# round 3, first look at k_step_pub: the ML test files, then the default bench line
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_assoc.py tests/test_gpu_random_worlds.py tests/test_gpu_new_landmarks.py -x -q -m gpu > gpurun_out/r3a_tests.txt 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r3a_tests.txt
timeout -k 10 400 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench_err.txt; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3a_bench.json'))
print('ms/step', d['ms_per_step'], 'kernel_ms', d.get('kernel_ms_per_step'))
r=d['roofline']; print(r['route'], r['frac'], r['avg_launch_ms'], r.get('frac_no_duplicates'))
print({k:r.get(k) for k in ('particles_sent_to_general_kernels_last_step','candidate_list_overflows_last_step')})
PY
