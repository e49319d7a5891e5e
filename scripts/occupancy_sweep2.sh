#!/bin/bash
# occupancy sweep for one bench mode: MODE="--gradient" or "--dtype f64";
# serial launches so that kernel times are isolated
MODE="$1"; shift
for occ in "$@"; do
GD_OCCUPANCY="$occ" python3 bench.py $MODE --steps 5 --warmup 2 --no-cpu-baseline --serial 2>&1 | tail -1 | OCC="$occ" python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['OCC'], round(d['value']/1e6,2), 'Mpairs/s', round(d['ms_per_step'],2), [(k['kernel'][4:], round(k['avg_ms'],3)) for k in d['kernels']])"
done
