"""Static instruction mix per barrier-delimited section of one kernel of an assembly listing (no GPU):
    python scripts/asm_sections.py gpurun_out/tmp/pub.s _ZN2pk10k_step_pubILi2ELi512EEEvNS_7PubArgsE"""
import re
import sys

path, name = sys.argv[1], sys.argv[2]
txt = open(path).read().split('\n')
i0 = next(k for k, l in enumerate(txt) if l.startswith(name + ':'))
sec, c = 0, {}


def show():
    print('section', sec, 'VALU', sum(v for k, v in c.items() if k.startswith('v')), dict(sorted(c.items())))


for l in txt[i0:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'):
        break
    if t.startswith('s_barrier'):
        show()
        sec += 1
        c = {}
        continue
    m = re.match(r'^([a-z_0-9]+)', t)
    if not m or t.endswith(':') or t.startswith(('.', ';')):
        continue
    op = m.group(1)
    if op.startswith('v_'):
        k = 'v_other'
        if 'f64' in op:
            k = 'v_f64'
        elif op.startswith(('v_cndmask', 'v_mov', 'v_accvgpr')):
            k = 'v_mov/cnd'
        elif op.startswith('v_cmp'):
            k = 'v_cmp'
    elif op.startswith('s_'):
        k = 's'
    elif op.startswith('ds_'):
        k = 'ds'
    elif op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
        k = 'g'
    else:
        k = 'other'
    c[k] = c.get(k, 0) + 1
show()
