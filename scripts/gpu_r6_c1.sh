# configs[1] on its new default route: rocprofv3 kernel stats of the default (k_step_pub<256 lanes>) and of k_step_fused (pub_small = 0), and the
# per-phase stamps of the default
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06; mkdir -p $O
for v in default fused; do
cd /tmp; rm -rf $R/gpurun_out/prof_c1
if [ $v = fused ]; then export PK_OPT_PUB_SMALL=0; else unset PK_OPT_PUB_SMALL; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c1 -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 120 --warmup 10 --particles 10000 --landmarks 500 > $R/$O/c1_bench_$v.json 2> $R/gpurun_out/prof_c1.log; echo "trace $v rc=$?"
cd $R
f=$(find gpurun_out/prof_c1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && (echo "# git $PK_GIT_SHA: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene --steps 120 --warmup 10 --particles 10000 --landmarks 500 (route: $v)"; head -14 "$f") > $O/kstats_c1_$v.csv
rm -rf gpurun_out/prof_c1
done
unset PK_OPT_PUB_SMALL
ST_WARM=15 ST_P=10240 ST_L=500 timeout -k 10 300 python scripts/gpu_stamps.py > $O/stamps_k_step_pub_256_lanes_10240x500.txt 2>&1; echo "stamps rc=$?"
head -12 $O/kstats_c1_default.csv | cut -c1-140; head -24 $O/stamps_k_step_pub_256_lanes_10240x500.txt
