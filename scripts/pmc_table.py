#!/usr/bin/env python3
"""Per-kernel table from the counter_collection.csv files of pmc_quick.sh."""
import glob
import sys
import pandas as pd

src = sys.argv[1]
frames = []
for f in glob.glob(f'{src}/*/*/*counter_collection.csv'):
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.startswith('mgk')]
    df['dur_us'] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    frames.append(df)
df = pd.concat(frames)
t = df.pivot_table(index='Kernel_Name', columns='Counter_Name',
                   values='Counter_Value', aggfunc='mean')
t['dur_us'] = df.groupby('Kernel_Name').dur_us.mean()
t['vgpr'] = df.groupby('Kernel_Name').VGPR_Count.first()
t['scratch'] = df.groupby('Kernel_Name').Scratch_Size.first()
cyc = t.dur_us * 2400.0                     # ~2.4 GHz
out = pd.DataFrame({
    'dur_us': t.dur_us.round(0),
    'vgpr': t.vgpr, 'scr': t.scratch,
    'waves/simd': (t.SQ_WAVE_CYCLES * 4 / (cyc * 1024)).round(2),
    'valu%': (100 * t.SQ_ACTIVE_INST_VALU * 4 / (cyc * 1024)).round(0),
    'lds%': (100 * t.SQ_LDS_IDX_ACTIVE / (cyc * 256)).round(0),
    'confl%lds': (100 * t.SQ_LDS_BANK_CONFLICT / t.SQ_LDS_IDX_ACTIVE).round(0),
    'wait%': (100 * t.SQ_WAIT_ANY / t.SQ_WAVE_CYCLES).round(0),
    'issue_wait%': (100 * t.SQ_WAIT_INST_ANY / t.SQ_WAVE_CYCLES).round(0),
    'VALU/w': (t.SQ_INSTS_VALU / t.SQ_WAVES).round(0),
    'SALU/w': (t.SQ_INSTS_SALU / t.SQ_WAVES).round(0),
    'LDS/w': (t.SQ_INSTS_LDS / t.SQ_WAVES).round(0),
    'ldscyc/w': (t.SQ_LDS_IDX_ACTIVE / t.SQ_WAVES).round(0),
})
pd.set_option('display.width', 250)
print(out.to_string())
