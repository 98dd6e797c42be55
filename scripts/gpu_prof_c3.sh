# kernel trace of the config 3 ML step (100 000 x 2 000)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_c3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --particles ${PMC_P:-100000} --landmarks ${PMC_L:-2000} > $R/gpurun_out/prof_c3.log 2>&1
cd $R
f=$(find gpurun_out/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_c3.csv && cut -c1-200 "$f" | head -12
