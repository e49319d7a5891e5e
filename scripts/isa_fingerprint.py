#!/usr/bin/env python3
"""Fingerprints of the gfx950 instruction streams of a representative set of
solver variants (hash of the ISA text without comments and directives): a
refactoring of the device headers that is meant to change NO generated code
is checked by running this before and after (round 6: the pruning of the
measured-and-dropped compile-time arms of mgk_oc.h).

    python scripts/isa_fingerprint.py out.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
DUMP = os.path.join(HERE, 'dump_isa.py')
CASES = {
    'f64_value_L16x4x4x1': '1 25 4 1 --f64 --oc=4 --layout=16x4x4x1 --tab',
    'f64_value_L16x4x4x3x1x1': '1 29 6 1 --f64 --oc=4 --layout=16x4x4x3x1x1 --tab',
    'f64_grad_L16x4x4x1': '1 25 4 2 --f64 --oc=4 --layout=16x4x4x1 --tab=2',
    'f32_value_L16x4x4x1': '1 25 4 1 --oc=4 --layout=16x4x4x1 --tab',
    'f32_grad_L16x4x4x1': '1 25 4 2 --oc=4 --layout=16x4x4x1 --tab=2',
    'f64_value_dyn_1_24_4': '1 24 4 1 --f64 --oc=4 --tab',
    'f64_grad_dyn_1_24_4': '1 24 4 2 --f64 --oc=4 --tab=2',
    'c2_f32_8_64_4': '8 64 4 1 --oc=8 --config2',
    'c2_f64_16_40_2': '16 40 2 1 --f64 --oc=8 --config2',
    'c2_f64_4_64_5': '4 64 5 1 --f64 --oc=8 --config2',
    'c2_f32_1_48_5': '1 48 5 1 --oc=8 --config2',
    'tang_f32_fly_4_0_2': '4 0 2 1 --oc=4 --tang',
    'tang_f64_fly_4_0_2': '4 0 2 1 --f64 --oc=4 --tang',
    'tang_f32_fly_grad': '4 0 2 2 --oc=4 --tang',
    'two_stage_f32_1_16_4': '1 16 4 1',
    'two_stage_f64_16_64_8': '16 64 8 1 --f64 --config2',
}


def fingerprint(item):
    name, args = item
    out = subprocess.run([sys.executable, DUMP, *args.split()],
                         capture_output=True, text=True)
    if out.returncode != 0:
        return name, 'ERROR ' + out.stderr[-300:]
    lines = []
    for line in out.stdout.splitlines():
        line = re.sub(r';.*$', '', line).strip()
        if not line or line.startswith('.') and not line.endswith(':'):
            continue
        lines.append(line)
    text = '\n'.join(lines)
    return name, f'{hashlib.sha256(text.encode()).hexdigest()[:16]} {len(lines)} lines'


def main():
    with ThreadPoolExecutor(max_workers=6) as ex:
        result = dict(ex.map(fingerprint, CASES.items()))
    for k, v in result.items():
        print(f'{k:28s} {v}')
    if len(sys.argv) > 1:
        with open(sys.argv[1], 'w') as f:
            json.dump(result, f, indent=1)


if __name__ == '__main__':
    main()
