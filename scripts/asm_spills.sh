#!/bin/bash
# Per barrier-delimited section of one kernel: VALU / LDS / VMEM counts plus scratch (spill) stores and loads (no GPU needed).
#   bash scripts/asm_spills.sh pk_k_observe_ml.hip _ZN2pk11k_step_regsENS_8RegsArgsE
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -mllvm -disable-machine-licm $ASM_FLAGS -S --cuda-device-only -o gpurun_out/tmp/k.s parakeet_slam_amd/csrc/$1 2>/dev/null
python3 - "$2" <<'PY'
import sys
name = sys.argv[1]
txt = open('gpurun_out/tmp/k.s').read().split('\n')
i0 = next(k for k, l in enumerate(txt) if l.startswith(name + ':'))
sec = 0
z = lambda: {'v': 0, 's': 0, 'ds': 0, 'g': 0, 'sc_st': 0, 'sc_ld': 0, 'rdlane': 0}
c = z(); tot = 0
for l in txt[i0:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'): break
    if t.startswith('s_barrier'):
        print('section', sec, c); sec += 1; c = z(); continue
    if t.startswith('scratch_store'): c['sc_st'] += 1
    elif t.startswith('scratch_load'): c['sc_ld'] += 1
    elif t.startswith(('v_readlane', 'v_writelane')): c['rdlane'] += 1
    elif t.startswith('v_'): c['v'] += 1; tot += 1
    elif t.startswith('s_'): c['s'] += 1
    elif t.startswith('ds_'): c['ds'] += 1
    elif t.startswith(('global_', 'buffer_', 'flat_')): c['g'] += 1
print('section', sec, c, 'total VALU', tot)
j = next(k for k, l in enumerate(txt) if l.strip().startswith('.name:') and l.strip().split()[-1] == name)
for l in txt[j:j + 14]:
    if any(w in l for w in ('vgpr_count', 'spill', 'private_segment_fixed_size', 'sgpr_count')): print(l.strip())
PY
