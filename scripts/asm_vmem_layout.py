"""Where the vector-memory instructions of a kernel section sit between its arithmetic (no GPU):
    python scripts/asm_vmem_layout.py file.s kernel section  ->  runs like  V312 ST5 V280 ST5 ..."""
import re
import sys

path, name, want = sys.argv[1], sys.argv[2], int(sys.argv[3])
txt = open(path).read().split('\n')
i0 = next(k for k, l in enumerate(txt) if l.startswith(name + ':'))
sec, runs = 0, []
for l in txt[i0:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'):
        break
    if t.startswith('s_barrier'):
        sec += 1
        continue
    if sec != want:
        continue
    k = None
    if t.startswith('v_'):
        k = 'V'
    elif t.startswith('global_store'):
        k = 'ST'
    elif t.startswith('global_load'):
        k = 'LD'
    elif t.startswith('ds_'):
        k = 'ds'
    elif t.startswith(('s_cbranch', 's_branch')):
        k = '|'
    if k is None:
        continue
    if runs and runs[-1][0] == k:
        runs[-1][1] += 1
    else:
        runs.append([k, 1])
print(' '.join('%s%d' % (k, n) if k != '|' else '|' for k, n in runs))
