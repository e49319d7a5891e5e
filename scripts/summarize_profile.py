#!/usr/bin/env python
"""Condense gpurun_out/prof (scripts/profile.sh) into the committed summaries
profiles/<tag>_kernel_stats.csv, <tag>_pmc.csv, <tag>_bench.json and
profiles/traffic.json (HBM bytes per launch from FETCH_SIZE / WRITE_SIZE,
which rocprofv3 reports in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B
request of a wide coalesced read -- MI355X_MICROARCH.md, HBM -- this solver's
reads are narrow gathers, so no x2 is applied; both figures are kept)."""
import glob
import json
import os
import shutil
import sys
import pandas as pd


def newest(pattern):
    """gpurun merges gpurun_out/ instead of replacing it: a directory may
    hold the files of several profiling runs -- take the latest."""
    return max(glob.glob(pattern), key=os.path.getmtime)


tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = 'gpurun_out/prof'
shutil.copy(newest(f'{src}/stats/*/*kernel_stats.csv'),
            f'profiles/{tag}_kernel_stats.csv')
shutil.copy(f'{src}/bench.json', f'profiles/{tag}_bench.json')
rows, traffic = [], {}
for d in ('pmc_a', 'pmc_b', 'pmc_fetch', 'pmc_write'):
    f = newest(f'{src}/{d}/*/*counter_collection.csv')
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.startswith('mgk')]
    df['dur_us'] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    g = df.groupby(['Kernel_Name', 'Counter_Name']).agg(
        value_per_dispatch=('Counter_Value', 'mean'),
        avg_dur_us=('dur_us', 'mean'),
        vgpr=('VGPR_Count', 'first'), lds=('LDS_Block_Size', 'first'),
        scratch=('Scratch_Size', 'first')).reset_index()
    rows.append(g)
    for _, r in g.iterrows():
        if r.Counter_Name in ('FETCH_SIZE', 'WRITE_SIZE'):
            traffic.setdefault(r.Kernel_Name, {})[r.Counter_Name + '_KiB'] = \
                float(r.value_per_dispatch)
pd.concat(rows).to_csv(f'profiles/{tag}_pmc.csv', index=False)
# the kernel trace's average duration per kernel (`bench.py --serial`: every
# launch alone on one stream) travels with the counters: bench.py prints it
# beside its own HIP-event duration (roofline.rocprof_avg_launch_ms)
stats = pd.read_csv(f'profiles/{tag}_kernel_stats.csv')
for _, r in stats.iterrows():
    if str(r['Name']).startswith('mgk') and r['Name'] in traffic:
        traffic[r['Name']]['rocprof_avg_launch_ms'] = float(r['AverageNs']) / 1e6
        traffic[r['Name']]['rocprof_calls'] = int(r['Calls'])
for k, v in traffic.items():
    v['hbm_bytes_per_launch'] = 1024 * (v.get('FETCH_SIZE_KiB', 0)
                                        + v.get('WRITE_SIZE_KiB', 0))
    v['hbm_bytes_per_launch_fetch_x2'] = 1024 * (
        2 * v.get('FETCH_SIZE_KiB', 0) + v.get('WRITE_SIZE_KiB', 0))
bench = json.loads([l for l in open(f'{src}/bench.json')
                    if l.startswith('{')][-1])
try:
    everything = json.load(open('profiles/traffic.json'))
    if 'kernels' in everything:          # pre-v7 layout: fp32 only
        everything = {'f32': everything}
except (OSError, ValueError):
    everything = {}
# key of this profile in traffic.json: the arithmetic, or argv[2] for the
# profiles of other workloads (gradient, configuration 2)
key = sys.argv[2] if len(sys.argv) > 2 else bench['dtype']
everything[key] = {'source': f'profiles/{tag}_pmc.csv',
                   'kernel_trace': f'profiles/{tag}_kernel_stats.csv',
                   'kernels': traffic}
json.dump(everything, open('profiles/traffic.json', 'w'), indent=1)
print(json.dumps(traffic, indent=1))
