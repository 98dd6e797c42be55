cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
for ps in 1 0; do
rm -rf $R/gpurun_out/prof_k
PK_OPT_PUB_SMALL=$ps timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_k -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-configs4 --no-refscene --no-probes --particles 10000 --landmarks 500 --steps 120 --warmup 10 > /dev/null 2> $R/gpurun_out/prof_k.log
f=$(find $R/gpurun_out/prof_k -name '*kernel_stats.csv' | head -1); cp $f $R/gpurun_out/r05/kstats_c1_ps$ps.csv
cut -c1-150 $f | head -14
done
