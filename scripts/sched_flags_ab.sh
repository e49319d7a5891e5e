#!/bin/bash
# Round 6: the headline step under alternative AMDGPU scheduling strategies
# (hipcc -mllvm flags through GD_HIPCC_EXTRA; the kernels are re-compiled on
# the box, the default last again).
cd $GRAFT_REPO_ROOT
for fl in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-schedule-metric-bias=50" "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule" ""; do
for dt in f64 f32; do
GD_HIPCC_EXTRA="$fl" timeout 900 python bench.py --dtype $dt --no-cpu-baseline --no-f32 --no-api --steps 200 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('[$fl] $dt:', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms', [(k['kernel'][8:34], round(k['isolated_ms'],3)) for k in d['kernels'][:3]])"
done; done
