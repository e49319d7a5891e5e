#!/usr/bin/env python3
"""The figures DESIGN.md quotes from profiles/<round>_*: step table and the
counter digest of the dominant launches.
    python scripts/r3_numbers.py [r03]"""
import json
import sys
import pandas as pd

rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
CLK = 2.4e9


def line(name):
    return json.loads([l for l in open(f'profiles/{rnd}_bench_{name}.json')
                       if l.startswith('{')][-1])


print('step table')
for name in ('f64', 'f32', 'grad64', 'grad32', 'c2', 'c2f64', 'tang'):
    d = line(name)
    a = d['step_aggregate']
    print(f"  {name:7s} {d['value'] / 1e6:7.1f} M  {d['ms_per_step']:6.2f} ms  "
          f"hbm {100 * a['hbm']['frac_step']:.1f} %  valu "
          f"{100 * a['compute']['frac_step']:.1f} %  lds "
          f"{100 * a['lds']['frac_step']:.1f} %   first call "
          f"{d.get('api_inclusive', {}).get('first_call_ms', 0):.1f} repeat "
          f"{d.get('api_inclusive', {}).get('repeat_call_ms', 0):.2f}")
    r = d['roofline']
    print(f"      dominant {r['kernel']}: {r['pairs_per_launch']} pairs in "
          f"{r['avg_launch_ms']:.3f} ms, algorithmic "
          f"{r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB -> "
          f"{r['achieved']:.0f} GB/s = {100 * r['frac']:.1f} %, measured "
          f"{r.get('achieved_measured_GBs') or 0:.0f} GB/s; valu "
          f"{100 * d['compute']['frac']:.1f} % lds {100 * d['lds']['frac']:.1f} %")
for name in ('gpr', 'gpr64', 'nws48'):
    d = line(name)
    print(f"  {name:7s} {d['value'] / 1e6:7.2f} M  {d['ms_per_step']:6.2f} ms")
    if name == 'nws48':
        print('     ', {k: v for k, v in d.items() if 'batch' in k or 'launch' in k})

print('counters (dominant launch of each profile)')
for name in ('f64', 'f32', 'grad64', 'grad32'):
    df = pd.read_csv(f'profiles/{rnd}_{name}_pmc.csv')
    df = df[df.Kernel_Name.str.startswith('mgk')]
    dur = df[df.Counter_Name == 'SQ_WAVES'].set_index('Kernel_Name')
    top = (dur.avg_dur_us).idxmax()
    k = df[df.Kernel_Name == top].set_index('Counter_Name')
    v = k.value_per_dispatch
    waves = v['SQ_WAVES']
    cyc_a = k.avg_dur_us['SQ_WAVES'] * 1e-6 * CLK
    cyc_b = k.avg_dur_us['SQ_ACTIVE_INST_VALU'] * 1e-6 * CLK
    print(f"  {name}: {top}  {k.avg_dur_us['SQ_WAVES']:.0f} us  vgpr "
          f"{k.vgpr.iloc[0]} scratch {k.scratch.iloc[0]}")
    print(f"      VALU busy {100 * 4 * v['SQ_ACTIVE_INST_VALU'] / (1024 * cyc_b):.0f} %"
          f"  LDS busy {100 * v['SQ_LDS_IDX_ACTIVE'] / (256 * cyc_b):.0f} %"
          f"  waves/SIMD {4 * v['SQ_WAVE_CYCLES'] / (1024 * cyc_a):.2f}"
          f"  conflicts/LDS-active {100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.0f} %")
    print(f"      per wave: VALU {v['SQ_INSTS_VALU'] / waves:.0f}  LDS "
          f"{v['SQ_INSTS_LDS'] / waves:.0f}  SALU {v['SQ_INSTS_SALU'] / waves:.0f}"
          f"  SMEM {v['SQ_INSTS_SMEM'] / waves:.0f}  VMEM {v['SQ_INSTS_VMEM'] / waves:.0f}"
          f"  LDS cycles {v['SQ_LDS_IDX_ACTIVE'] / waves:.0f}")
t = json.load(open('profiles/traffic.json'))
for key in ('f64', 'f32'):
    ks = t[key]['kernels']
    top = max(ks, key=lambda k_: ks[k_]['hbm_bytes_per_launch'])
    print(f"  traffic {key}: {top} {ks[top]['hbm_bytes_per_launch_fetch_x2'] / 1e6:.0f} MB "
          f"(2 x FETCH + WRITE), step total "
          f"{sum(x['hbm_bytes_per_launch_fetch_x2'] for x in ks.values()) / 1e6:.0f} MB")
