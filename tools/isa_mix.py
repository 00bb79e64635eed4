#!/usr/bin/env python
"""Instruction mix between consecutive s_barrier's of a kernel in a hipcc -save-temps .s file (one loop tile per barrier in the
attention kernels): python tools/isa_mix.py file.s kernel_substring [mfma_count_filter]"""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('buffer_', 'global_', 'flat_')): return 'VMEM'
    if op.startswith('s_waitcnt'): return 'WAIT'
    if op.startswith('s_nop'): return 'NOP'
    if op.startswith('s_barrier'): return 'BAR'
    if op.startswith(('s_load', 's_buffer')): return 'SMEM'
    if op.startswith('s_'): return 'SALU'
    if op.startswith('v_accvgpr'): return 'ACC'
    if op.startswith('v_'): return 'VALU'
    return 'OTHER'


def main():
    path, sub = sys.argv[1], sys.argv[2]
    want = int(sys.argv[3]) if len(sys.argv) > 3 else None
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l)] + [len(lines)]
    for a, b in zip(starts[:-1], starts[1:]):
        if sub not in lines[a]:
            continue
        print(lines[a].split(':')[0])
        bars = [i for i in range(a, b) if lines[i].strip().startswith('s_barrier')]
        for x, y in zip(bars[:-1], bars[1:]):
            c, ops = collections.Counter(), collections.Counter()
            for l in lines[x:y]:
                t = l.strip()
                if not t or t.startswith((';', '.')) or t.endswith(':'):
                    continue
                op = t.split()[0]
                k = classify(op)
                c[k] += 1
                if k in ('VALU', 'SALU'):
                    ops[op] += 1
            if want is None or c['MFMA'] == want:
                print(f'  lines {x}-{y}:', dict(c), 'issued', sum(v for k, v in c.items() if k not in ('WAIT',)))
                print('     ', ops.most_common(40))


if __name__ == '__main__':
    main()
