# kernel trace of the default ML step (100 000 x 2 000): only the timed filter
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_r3b
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3b -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-probes --steps 6 --warmup 2 > $R/gpurun_out/prof_r3b.log 2>&1
cd $R
f=$(find gpurun_out/prof_r3b -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3b_kernel_stats.csv && cut -c1-220 "$f" | head -24
rm -rf gpurun_out/prof_r3b
