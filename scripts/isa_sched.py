#!/usr/bin/env python3
"""Print a compact instruction-class string of a kernel's MFMA region from a hipcc -S file.
usage: isa_sched.py file.s <substring of mangled kernel name>"""
import sys
txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith('_ZN') and key in l and ':' in l)
end = next(i for i in range(start, len(txt)) if '.end_amdhsa_kernel' in txt[i] or txt[i].startswith('.Lfunc_end'))
lines = [l.strip() for l in txt[start:end] if l.strip() and not l.strip().startswith(';')]
idx = [i for i, l in enumerate(lines) if 'v_mfma' in l]
print('mfma count', len(idx), 'instructions', len(lines))
seq = []
for l in lines[max(0, idx[0] - 80): idx[-1] + 5]:
    op = l.split()[0]
    if op.startswith('v_mfma'): seq.append('M')
    elif op.startswith('ds_read'): seq.append('r')
    elif op.startswith('ds_write'): seq.append('w')
    elif op.startswith('s_waitcnt'): seq.append('W[' + l.split(None, 1)[1].replace(' ', '') + ']')
    elif op.startswith('s_barrier'): seq.append('BAR')
    elif op.startswith('global_load_lds'): seq.append('G')
    elif op.startswith('global_load') or op.startswith('buffer_load'): seq.append('L')
    elif op.startswith('global_store'): seq.append('S')
    elif op.startswith('s_cbranch'): seq.append('BR')
    elif l.endswith(':'): seq.append('\n' + l)
    else: seq.append('.')
print(''.join(seq))
