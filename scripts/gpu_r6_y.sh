# the flag launch folded into k_cand_entries: the whole GPU suite, then configs[1] and configs[2] once each
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/ -q -m gpu > $O/suite.log 2>&1; rc=$?; tail -4 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/suite.log 2>&1 || rc=1
if grep -q "Memory access fault" $O/suite.log; then echo "FAULT"; exit 1; fi
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 120 --warmup 10 --particles 10000 --landmarks 500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs1 ms/step %.4f observe %.4f route %s' % (d['ms_per_step'], d['kernel_ms_per_step']['observe'], d['roofline']['route']))"
done
