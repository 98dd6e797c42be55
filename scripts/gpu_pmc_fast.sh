# Instruction mix of the ML-step kernels (k_assoc_grid hand-off, k_observe_fast) from SQ counters.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${PMC_P:-10000}; L=${PMC_L:-500}
mkdir -p $R/gpurun_out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --particles $P --landmarks $L --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_SQ*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        if 'assoc' in k or 'observe' in k:
            print(k, {a: '%.3g' % (b / n[(k, a)]) for a, b in v.items()})
PY
