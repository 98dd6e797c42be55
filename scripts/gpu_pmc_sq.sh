# SQ instruction mix / wait counters of the ML-step kernels at PMC_P x PMC_L (default 51 200 x 2 000: 200 particles
# per CU), one counter set per rocprofv3 pass.  Writes gpurun_out/pmc_sq.json.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${PMC_P:-51200}; L=${PMC_L:-2000}
mkdir -p $R/gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcsq_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcsq_$i -- python3 $R/bench.py --steps ${SQ_STEPS:-3} --warmup ${SQ_WARMUP:-1} --particles $P --landmarks $L --no-cpu-baseline --no-secondary --no-probes ${PMC_ARGS} > $R/gpurun_out/pmcsq_$i.log 2>&1
  tail -1 $R/gpurun_out/pmcsq_$i.log | cut -c1-160
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/pmcsq_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        if any(w in k for w in ('assoc', 'observe', 'step_')):
            out.setdefault(k, {}).update({a: b / n[(k, a)] for a, b in v.items()})
for k, v in out.items():
    print(k, {a: '%.4g' % b for a, b in v.items()})
json.dump(out, open('gpurun_out/pmc_sq.json', 'w'), indent=1)
PY
