#!/usr/bin/env python3
"""The round-4 tables of DESIGN.md from profiles/: bench lines, the counter
digest per kernel (profiles/r04_<name>_pmc.csv, formulas of
scripts/pmc_table.py) and the HBM traffic against the algorithmic bytes.

    python scripts/r4_numbers.py [r04] [f64 f32 grad64 ...]
"""
import json
import os
import sys
import pandas as pd

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r04'
names = sys.argv[2:] or ['f64', 'f32', 'grad64', 'grad32', 'c2', 'c2f64',
                         'tang', 'tanggrad']
pd.set_option('display.width', 250)
traffic = json.load(open(os.path.join(root, 'traffic.json')))
for name in names:
    try:
        line = open(os.path.join(root, f'{rnd}_bench_{name}.json')).read()
        b = json.loads(line.strip().splitlines()[-1])
    except OSError:
        b = json.load(open(os.path.join(root, f'{rnd}_{name}_bench.json')))
    sa = b['step_aggregate']
    print(f'== {name}: {b["value"] / 1e6:.2f} M pairs/s, {b["ms_per_step"]:.3f} ms'
          f' | step models: HBM {100 * sa["hbm"]["frac_step"]:.1f} %,'
          f' VALU {100 * sa["compute"]["frac_step"]:.1f} %,'
          f' LDS {100 * sa["lds"]["frac_step"]:.1f} %'
          f' | iterations {b.get("mean_cg_iterations")}')
    df = pd.read_csv(os.path.join(root, f'{rnd}_{name}_pmc.csv'))
    t = df.pivot_table(index='Kernel_Name', columns='Counter_Name',
                       values='value_per_dispatch', aggfunc='mean')
    meta = df.groupby('Kernel_Name').agg(dur_us=('avg_dur_us', 'mean'),
                                         vgpr=('vgpr', 'first'),
                                         scr=('scratch', 'first'))
    cyc = meta.dur_us * 2400.0
    out = pd.DataFrame({
        'dur_us': meta.dur_us.round(0), 'vgpr': meta.vgpr, 'scr': meta.scr,
        'waves/simd': (t.SQ_WAVE_CYCLES * 4 / (cyc * 1024)).round(2),
        'valu%': (100 * t.SQ_ACTIVE_INST_VALU * 4 / (cyc * 1024)).round(0),
        'lds%': (100 * t.SQ_LDS_IDX_ACTIVE / (cyc * 256)).round(0),
        'confl%lds': (100 * t.SQ_LDS_BANK_CONFLICT
                      / t.SQ_LDS_IDX_ACTIVE).round(0),
        'VALU/w': (t.SQ_INSTS_VALU / t.SQ_WAVES).round(0),
        'LDS/w': (t.SQ_INSTS_LDS / t.SQ_WAVES).round(0),
    })
    alg = {k['kernel']: k['algorithmic_bytes'] for k in b['kernels']}
    tr = traffic.get(name, {}).get('kernels', {})
    out['HBM MB'] = [round(tr.get(k, {}).get('hbm_bytes_per_launch_fetch_x2',
                                             float('nan')) / 1e6, 1)
                     for k in out.index]
    out['x algorithmic'] = [round(tr.get(k, {}).get(
        'hbm_bytes_per_launch_fetch_x2', float('nan')) / alg[k], 2)
        if k in alg else float('nan') for k in out.index]
    print(out[out.index.str.startswith('mgk') & ~out.index.str.contains('tables')].to_string())
