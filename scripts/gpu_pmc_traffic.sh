# HBM traffic of the dominant kernel from PMC counters (MI355X_MICROARCH.md section HBM):
# FETCH_SIZE and WRITE_SIZE in separate passes; calibration on k_copy_slots (known byte count).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cat > /tmp/traffic_run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import bench
from parakeet_slam_amd import _lib
P, L = 10000, 500
means, covs, scans = bench.synthetic_inputs(L, 6)
f = _lib.DeviceFilter(P, L)
f.upload_map(means, covs.reshape(L, 25))
ids = np.arange(1, L + 1, dtype=np.int32)
for s in range(3):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, ids=ids, domain=1)   # k_observe<true,2>
for s in range(3, 6):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, domain=1)            # k_step_fused
f.set_option("fused_step", 0)
for s in range(3, 6):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, domain=1)            # k_assoc_grid hand-off + k_observe_fast
f.download_landmarks(0, 1)                                                     # k_copy_slots: P slots copied
f.synchronize()
print("slot_bytes", f.particle_bytes() - 32)
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 /tmp/traffic_run.py > $R/gpurun_out/pmc_$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_$c.log
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            key = 'observe_known' if ('k_observe<true' in k or 'k_observe_single' in k) else 'step_fused' if 'k_step_fused' in k else 'observe_ml' if 'k_observe_fast' in k else 'assoc_grid' if ('k_assoc_grid' in k and ', false, ' in k) else 'copy_slots' if 'k_copy_slots' in k else None
            if key: res[key][c].append(float(r['Counter_Value']))
out = {}
for k, v in res.items():
    out[k] = {c: (sum(x) / len(x), len(x)) for c, x in v.items()}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmc_traffic_raw.json', 'w'), indent=1)
PY
