# HBM traffic of the dominant kernels at PMC_P x PMC_L (default BASELINE configs[2], 100 000 x 2 000) from PMC
# counters (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (KiB units;
# FETCH_SIZE doubled per the gfx950 note; WRITE_SIZE calibrated on k_copy_slots' known byte count).
# Writes gpurun_out/pmc_traffic_${P}x${L}.json (copy to profiles/rNN/ to have bench.py report it).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${PMC_P:-100000}; L=${PMC_L:-2000}
mkdir -p $R/gpurun_out
cat > /tmp/traffic_run.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import bench
from parakeet_slam_amd import _lib
P, L = $P, $L
means, covs, scans = bench.synthetic_inputs(L, 8)
f = _lib.DeviceFilter(P, L)
f.upload_map(means, covs.reshape(L, 25))
ids = np.arange(1, L + 1, dtype=np.int32)
for s in range(3):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, ids=ids, domain=1)   # supplied ids
for s in range(3, 6):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, domain=1)            # production ML route
if 512 < L <= 2048:
    f.set_option("pub_step", 0)
    for s in range(3, 6):
        f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, domain=1)        # k_step_regs instead of k_step_pub
    f.set_option("pub_step", 1)
if L > 2048:
    f.set_option("pub_step", 0)                                                # the two-kernel route instead of k_step_pub_big
elif L > 512:
    f.set_option("regs_step", 0)
else:
    f.set_option("fused_step", 0)
for s in range(3, 6):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, domain=1)            # two-kernel ML route
f.download_landmarks(0, 1)                                                     # k_copy_slots: P slots copied
f.synchronize()
print("slot_bytes", f.particle_bytes() - 48)
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 /tmp/traffic_run.py > $R/gpurun_out/pmc_$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_$c.log
done
cd $R
PMC_P=$P PMC_L=$L python3 - <<'PY'
import csv, glob, collections, json, os
P, L = int(os.environ["PMC_P"]), int(os.environ["PMC_L"])
res = collections.defaultdict(lambda: collections.defaultdict(list))
def key_of(k):
    if 'k_observe<true' in k or 'k_observe_single' in k: return 'observe_known'
    if 'k_step_fused' in k: return 'step_fused'
    if 'k_step_pub_big' in k: return 'step_pub_big'
    if 'k_step_pub' in k: return 'step_pub'
    if 'k_step_regs' in k: return 'step_regs'
    if 'k_observe_fast' in k: return 'observe_ml'
    if 'k_observe_sweep' in k: return 'observe_sweep'
    if 'k_assoc_grid' in k and ', false, ' in k: return 'assoc_grid'
    if 'k_copy_slots' in k: return 'copy_slots'
    return None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            key = key_of(r['Kernel_Name'])
            if key: res[key][c].append(float(r['Counter_Value']))
slot = None
for line in open('gpurun_out/pmc_WRITE_SIZE.log'):
    if line.startswith('slot_bytes'): slot = int(line.split()[1])
raw, out = {}, {}
for k, v in res.items():
    # launches that return at once (the stand-by instance of a kernel pair, the general kernels with nothing flagged) move
    # next to nothing: they are not averaged in
    v = {c: [y for y in x if y > 0.01 * max(x)] or x for c, x in v.items()}
    raw[k] = {c: (sum(x) / len(x), len(x)) for c, x in v.items()}
    if 'FETCH_SIZE' in raw[k] and 'WRITE_SIZE' in raw[k]:
        out[k] = (2.0 * raw[k]['FETCH_SIZE'][0] + raw[k]['WRITE_SIZE'][0]) * 1024.0
doc = {"git": os.environ.get("PK_GIT_SHA", "unknown"), "config": {"particles": P, "landmarks": L, "blobs": L},
       "what": "HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 --pmc, one counter per pass, "
               "gfx950 corrections of MI355X_MICROARCH.md); FETCH_SIZE counts reads served by the Infinity Cache too",
       "algorithmic_bytes_per_launch": P * L * 224, "slot_bytes": slot,
       "copy_slots_expected_write_bytes": (P * slot) if slot else None,
       "bytes_per_launch": out, "raw_kib_mean_and_launches": raw}
name = 'gpurun_out/pmc_traffic_%dx%d.json' % (P, L)
json.dump(doc, open(name, 'w'), indent=1)
print(json.dumps({k: round(v / (P * L * 224.0), 3) for k, v in out.items()}), "x algorithmic ->", name)
PY
