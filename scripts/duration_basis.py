#!/usr/bin/env python3
"""Per kernel, the two duration bases side by side: the isolated HIP-event
duration `bench.py` quotes (kernels[].isolated_ms: the launch alone on one
stream, after the timed region) and the rocprofv3 --kernel-trace --stats
average of the same command run with --serial (profiles/<tag>_kernel_stats.csv)
-- they come from different processes and must agree.

    python scripts/duration_basis.py r04 [f64 f32 grad64 ...]
"""
import csv
import json
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r04'
names = sys.argv[2:] or ['f64', 'f32', 'grad64', 'grad32', 'c2', 'c2f64',
                         'tang', 'tanggrad']
worst = 0.0
for name in names:
    try:
        bench = json.load(open(os.path.join(root, f'{rnd}_{name}_bench.json')))
        stats = {r['Name']: float(r['AverageNs']) * 1e-6 for r in csv.DictReader(
            open(os.path.join(root, f'{rnd}_{name}_kernel_stats.csv')))}
    except OSError:
        continue
    print(f'== {rnd}_{name}: {bench["value"] / 1e6:.2f} M pairs/s, '
          f'{bench["ms_per_step"]:.3f} ms per step')
    print(f'{"kernel":52s} {"pairs":>8s} {"HIP events ms":>14s} '
          f'{"rocprof ms":>11s} {"ratio":>6s}')
    for k in bench['kernels']:
        iso, prof = k.get('isolated_ms'), stats.get(k['kernel'])
        if iso is None or prof is None:
            continue
        ratio = iso / prof
        if k['isolated_ms'] > 0.05 * bench['ms_per_step']:
            worst = max(worst, abs(ratio - 1))
        print(f'{k["kernel"]:52s} {k["pairs"]:8d} {iso:14.4f} {prof:11.4f} '
              f'{ratio:6.3f}')
print(f'largest deviation among launches above 5 % of their step: '
      f'{100 * worst:.1f} %')
