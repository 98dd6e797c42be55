# HBM-side traffic of the one-pass ML kernel OVER THE DRIVER'S WINDOW (bench.py --steps 20 --warmup 5: the mean over its 25 launches),
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (units and gfx950 correction as in gpu_pmc_traffic2.sh, whose passes look at
# three early steps of an artificial trajectory).  PMC_P x PMC_L (default 100 000 x 2 000).  Writes gpurun_out/pmc_traffic_window_PxL.json.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${PMC_P:-100000}; L=${PMC_L:-2000}
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcw_$c
  timeout 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcw_$c -- python3 $R/bench.py --steps 20 --warmup 5 --particles $P --landmarks $L --no-cpu-baseline --no-secondary --no-probes --no-configs4 --no-refscene > $R/gpurun_out/pmcw_$c.log 2>&1
  tail -1 $R/gpurun_out/pmcw_$c.log | cut -c1-120
done
cd $R
PMC_P=$P PMC_L=$L python3 - <<'PY'
import csv, glob, collections, json, os
P, L = int(os.environ["PMC_P"]), int(os.environ["PMC_L"])
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob('gpurun_out/pmcw_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c and 'k_step_' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('(')[0].replace('void pk::', '')
                acc[k] += float(r['Counter_Value']); n[k] += 1
    for k in acc:
        raw.setdefault(k, {})[c] = [acc[k] / n[k], n[k]]
out = {k: (2.0 * v['FETCH_SIZE'][0] + v['WRITE_SIZE'][0]) * 1024.0 for k, v in raw.items() if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v and v['FETCH_SIZE'][1] >= 20}
doc = {"git": os.environ.get("PK_GIT_SHA", "unknown"), "config": {"particles": P, "landmarks": L, "blobs": L},
       "what": "HBM-side bytes per launch of the one-pass ML kernel, MEAN OVER THE 25 LAUNCHES of bench.py --steps 20 --warmup 5 (the driver's window and its warm-up) = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 --pmc, one counter per pass); FETCH_SIZE counts reads served by the Infinity Cache too, not reads served by an XCD's L2",
       "algorithmic_bytes_per_launch": P * L * 224, "bytes_per_launch": out, "x_algorithmic": {k: v / (P * L * 224.0) for k, v in out.items()}, "raw_kib_mean_and_launches": raw}
name = 'gpurun_out/pmc_traffic_window_%dx%d.json' % (P, L)
json.dump(doc, open(name, 'w'), indent=1)
print(json.dumps({k: round(v, 3) for k, v in doc["x_algorithmic"].items()}), "x algorithmic ->", name)
PY
