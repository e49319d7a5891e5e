#!/usr/bin/env python3
"""Per-basic-block instruction mix of a `hipcc -S` listing (see dump_isa.py):
VALU / SALU / LDS / VMEM / SMEM / branch counts, to find what the CG loop
spends its issue slots on."""
import re
import sys
from collections import Counter

def kind(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_load') or op.startswith('s_buffer'):
        return 'smem'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith(('s_waitcnt', 's_nop', 's_barrier')):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    return 'other'

blocks, cur = [], None
for line in open(sys.argv[1]):
    m = re.match(r'^(\.LBB\d+_\d+|[A-Za-z_][\w.]*):', line)
    if m:
        cur = [m.group(1), Counter(), Counter(), line.strip()]
        blocks.append(cur)
        continue
    m = re.match(r'^\s+([a-z_0-9]+)\s', line)
    if m and cur is not None and not line.strip().startswith(('.', ';')):
        op = m.group(1)
        cur[1][kind(op)] += 1
        cur[2][op] += 1
verbose = len(sys.argv) > 2
for name, kinds, ops, _ in blocks:
    n = sum(kinds.values())
    if n < 8:
        continue
    print(f'{name:12s} n={n:4d} ' + ' '.join(f'{k}={v}' for k, v in
                                              sorted(kinds.items())))
    if verbose and name in sys.argv[2:]:
        for op, c in ops.most_common():
            print(f'      {op:28s} {c}')
