#!/usr/bin/env python3
"""instruction-class counts of every loop of a kernel in a `hipcc -S --cuda-device-only` file, plus its register
budget: isa_loops.py file.s <substring of the mangled kernel name> [...]"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().split('\n')


def dump(key):
    start = next(i for i, l in enumerate(txt) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
    end = next(i for i in range(start, len(txt)) if '.end_amdhsa_kernel' in txt[i])
    body_end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
    lines = [l.strip() for l in txt[start:body_end] if l.strip() and not l.strip().startswith(';')]
    meta = {m.group(1): m.group(2) for l in txt[body_end:end] for m in [re.search(r'\.amdhsa_(next_free_vgpr|accum_offset|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size)\s+(\S+)', l)] if m}
    print(key, 'instructions', len(lines), meta)
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
    for i, l in enumerate(lines):
        m = re.match(r's_cbranch\w+\s+(\S+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            c = Counter()
            for b in lines[labels[m.group(1)]:i]:
                op = b.split()[0]
                if op.startswith('v_mfma'): c['mfma'] += 1
                elif op.startswith('v_'): c['valu'] += 1
                elif op.startswith('s_waitcnt'): c['wait'] += 1
                elif op.startswith('s_'): c['salu'] += 1
                elif 'load' in op and not op.startswith('ds_'): c['load'] += 1
                elif 'store' in op and not op.startswith('ds_'): c['store'] += 1
                elif op.startswith('ds_'): c['lds'] += 1
                else: c['other'] += 1
            print('  loop', m.group(1), i - labels[m.group(1)], dict(c))


for k in sys.argv[2:]:
    dump(k)
