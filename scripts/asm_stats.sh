#!/bin/bash
# Static instruction statistics of one kernel of one translation unit (no GPU needed):
#   bash scripts/asm_stats.sh pk_k_observe_ml.hip _ZN2pk12k_step_fusedILb1EEEvNS_9FusedArgsE
# prints registers / scratch and VALU, SALU, LDS, VMEM counts per barrier-delimited section.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tmp
src=parakeet_slam_amd/csrc/$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $ASM_FLAGS -S --cuda-device-only -o gpurun_out/tmp/k.s "$src" 2>/dev/null
python3 - "$2" <<'PY'
import re, sys
name = sys.argv[1]
txt = open('gpurun_out/tmp/k.s').read().split('\n')
i0 = next(k for k, l in enumerate(txt) if l.startswith(name + ':'))
sec, c, tot = 0, {'v': 0, 's': 0, 'ds': 0, 'g': 0}, 0
for l in txt[i0:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'):
        break
    if t.startswith('s_barrier'):
        print('section', sec, c); sec += 1; c = {'v': 0, 's': 0, 'ds': 0, 'g': 0}; continue
    if t.startswith('v_'): c['v'] += 1; tot += 1
    elif t.startswith('s_'): c['s'] += 1
    elif t.startswith('ds_'): c['ds'] += 1
    elif t.startswith(('global_', 'scratch_', 'buffer_', 'flat_')): c['g'] += 1
print('section', sec, c, 'total VALU', tot)
j = next(k for k, l in enumerate(txt) if l.strip().startswith('.name:') and l.strip().split()[-1] == name)
for l in txt[j:j + 12]:
    if any(w in l for w in ('vgpr_count', 'spill', 'private_segment_fixed_size', 'sgpr_count')): print(l.strip())
PY
