# Vector-memory path counters (texture addresser TA, vector L1 TCP) of the ML-step kernels at PMC_P x PMC_L over bench.py's own window
# (warm-up 5, 20 steps), one counter set per rocprofv3 pass; PMC_TAG names the output, PMC_ENV is put in the environment (PK_OPT_PUB_DUO=0).
# Writes gpurun_out/r06/pmc_ta_<tag>.json.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${PMC_P:-20000}; L=${PMC_L:-5000}; TAG=${PMC_TAG:-x}
mkdir -p $R/gpurun_out/r06
[ -n "$PMC_ENV" ] && export $PMC_ENV
i=0
for set in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcta_${TAG}_$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcta_${TAG}_$i -- python3 $R/bench.py --steps 20 --warmup 5 --particles $P --landmarks $L --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene > $R/gpurun_out/pmcta_${TAG}_$i.log 2>&1
  echo "pass $i ($set): rc $?"
done
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob('gpurun_out/pmcta_%s_*/**/*counter_collection.csv' % tag, recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:48]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        if any(w in k for w in ('step_pub', 'observe', 'step_')):
            out.setdefault(k, {}).update({a: b / n[(k, a)] for a, b in v.items()})
            out[k]['launches'] = max(out[k].get('launches', 0), max(n[(k, a)] for a in v))
for k, v in out.items():
    print(k, {a: '%.4g' % b for a, b in v.items()})
json.dump(out, open('gpurun_out/r06/pmc_ta_%s.json' % tag, 'w'), indent=1)
PY
