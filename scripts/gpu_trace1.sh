# kernel timeline of the default bench (for gap analysis)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -- python3 $R/bench.py --no-cpu-baseline --steps 30 ${TL_ARGS} > $R/gpurun_out/prof_tl.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
fs=sorted(glob.glob('gpurun_out/prof_tl/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows=list(csv.DictReader(open(fs[-1])))
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:44]) for r in rows)
idx=[i for i,e in enumerate(ev) if 'k_motion' in e[2]]
i0=idx[20]; i1=idx[21]; t0=ev[i0][0]
for e in ev[i0:i1+1]:
    print('%8.2f %8.2f  dur %7.2f  %s' % ((e[0]-t0)/1e3,(e[1]-t0)/1e3,(e[1]-e[0])/1e3,e[2]))
PY
