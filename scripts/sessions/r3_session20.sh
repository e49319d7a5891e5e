#!/bin/bash
# round 3, session 20: gathers in flight (GD_OC_GCH) and setup chunk (GD_OC_CHUNK) after the setup changes
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for fl in "" "-DGD_OC_GCH=4" "-DGD_OC_GCH=12" "-DGD_OC_GCH=16" "-DGD_OC_CHUNK=2" "-DGD_OC_CHUNK=8"; do
  for dt in f64 f32; do
    GD_HIPCC_EXTRA="$fl" timeout 900 python bench.py --dtype $dt --no-cpu-baseline --no-api > gpurun_out/ab.json 2> gpurun_out/ab.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab.json').read().strip().split('\n')[-1]); print('$fl', '$dt', round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4))"
  done
done
