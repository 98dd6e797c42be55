# the whole -m gpu suite as the driver runs it (without -x: every failure shows), then smoke()
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > $O/suite.log 2>&1; rc=$?
python -c "import __graft_entry__ as g; g.smoke()" >> $O/suite.log 2>&1 || rc=1
echo "suite rc $rc" | tee -a $O/suite.log
tail -15 $O/suite.log
if grep -q "Memory access fault" $O/suite.log; then echo "FAULT"; exit 1; fi
exit $rc
