#!/usr/bin/env python3
"""Instruction counts per phase of one owner-computes solver: compiles the
variant with -DGD_MARKS (mgk_oc.h GD_MARK) through scripts/dump_isa.py and
counts VALU / SALU / LDS / VMEM instructions between the marks, in program
order of the ISA.

    python scripts/isa_phases.py W S R [C] [--f64] [--layout=16x4x4x1] [--tab=2]
"""
import collections
import os
import re
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
env = dict(os.environ)
env['GD_HIPCC_EXTRA'] = (env.get('GD_HIPCC_EXTRA', '') + ' -DGD_MARKS').strip()
isa = subprocess.run([sys.executable, os.path.join(here, 'dump_isa.py')]
                     + sys.argv[1:], env=env, capture_output=True, text=True,
                     check=True).stdout
phase = 'prologue'
counts = collections.OrderedDict()
for line in isa.split('\n'):
    m = re.search(r'GDMARK (\w+)', line)
    if m:
        phase = m.group(1)
        continue
    m = re.match(r'\s+([a-z_0-9]+)\s', line)
    if not m or line.lstrip().startswith(('.', ';')):
        continue
    op = m.group(1)
    kind = ('valu' if op.startswith('v_') else
            'salu' if op.startswith('s_') else
            'lds' if op.startswith('ds_') else
            'vmem' if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_'))
            else None)
    if kind is None:
        continue
    counts.setdefault(phase, collections.Counter())[kind] += 1
print(f'{"phase":12s} {"valu":>6s} {"salu":>6s} {"lds":>6s} {"vmem":>6s}')
for ph, c in counts.items():
    print(f'{ph:12s} {c["valu"]:6d} {c["salu"]:6d} {c["lds"]:6d} {c["vmem"]:6d}')
