# full GPU suite, then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/k_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/k_pytest.log; [ $rc -eq 0 ] && \
timeout -k 10 400 python bench.py > gpurun_out/k_bench.json 2> gpurun_out/k_bench.err; echo "rc $?"; cut -c1-1500 gpurun_out/k_bench.json
