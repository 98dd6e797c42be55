#!/bin/bash
# the default bench line as the driver runs it (validates the line's round-6 fields), and the one duo test whose assertion changed
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/m_bench.json 2> $O/m_bench.err; echo "bench rc $?"
tail -c 600 $O/m_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/m_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.3f frac %.3f range %s traffic %s window %s" % (d["value"], d["ms_per_step"], r["frac"], r.get("frac_range_over_steps"), r.get("traffic"), r.get("traffic_window")))
print("replayed_from", r.get("replayed_from"), "code", r.get("kernel_code"))
print("warmup", d.get("warmup_steps"))
print("all_steps", d["per_step"]["all_steps"], d["per_step"]["kernel_ms_min_max_timed_window"])
s4 = d.get("configs4_shard", {})
print("shard", s4.get("ms_per_step"), s4.get("roofline", {}).get("frac"), s4.get("warmup_steps"), s4.get("late_window", {}).get("ms_per_step"), s4.get("error"))
c1 = d.get("configs1", {})
print("configs1", c1.get("ms_per_step"), c1.get("roofline", {}).get("frac"), c1.get("k_step_fused", {}).get("ms_per_step"))
print("grow", json.dumps(d.get("refscene", {}).get("new_landmarks"))[:900])
print("grow at scale", json.dumps(d.get("refscene", {}).get("new_landmarks_at_scale"))[:900])
print("cpu", d.get("cpu_baseline"))
PY
timeout -k 10 300 python -m pytest tests/test_gpu_duo.py -q -x -m gpu -k "two_instances" 2>&1 | tail -3
