#!/bin/bash
# round 3, session 21: gathers in flight in double, values and value + gradient
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for fl in "-DGD_OC_GCH=16" "-DGD_OC_GCH=20" "-DGD_OC_GCH=25" "-DGD_OC_GCH=32"; do
    GD_HIPCC_EXTRA="$fl" timeout 900 python bench.py --dtype f64 --no-cpu-baseline --no-api > gpurun_out/ab.json 2> gpurun_out/ab.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab.json').read().strip().split('\n')[-1]); print('$fl', 'f64', round(d['value']/1e6,1), round(d['ms_per_step'],3), [round(k['isolated_ms'],3) for k in d['kernels']])"
done
for fl in "" "-DGD_OC_GCH=4" "-DGD_OC_GCH=12" "-DGD_OC_GCH=16"; do
  for dt in f64 f32; do
    GD_HIPCC_EXTRA="$fl" timeout 900 python bench.py --gradient --dtype $dt --no-cpu-baseline --no-api > gpurun_out/ab.json 2> gpurun_out/ab.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab.json').read().strip().split('\n')[-1]); print('$fl', 'grad $dt', round(d['value']/1e6,1), round(d['ms_per_step'],3))"
  done
done
for fl in "" "-DGD_OC_GCH=16"; do
  for dt in f64 f32; do
    GD_HIPCC_EXTRA="$fl" timeout 900 python bench.py --config 2 --dtype $dt --no-cpu-baseline --no-api > gpurun_out/ab.json 2> gpurun_out/ab.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab.json').read().strip().split('\n')[-1]); print('$fl', 'config 2 $dt', round(d['value']/1e6,2), round(d['ms_per_step'],3))"
  done
done
