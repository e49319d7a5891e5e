#!/bin/bash
# per-variant occupancy sweep (serial launches so kernel times are isolated)
for occ in "1:12:6,1:16:5,1:20:4,1:24:4,1:28:3,1:32:3" "1:12:5,1:16:4,1:20:3,1:24:3,1:28:2,1:32:2" "1:12:4,1:16:3,1:20:5,1:24:5,1:28:4,1:32:4" "1:12:8,1:16:6,1:20:2,1:24:2,1:28:2,1:32:2"; do
GD_OCCUPANCY="$occ" python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial 2>&1 | tail -1 | OCC="$occ" python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['OCC'], round(d['value']/1e6,2), 'Mpairs/s', round(d['ms_per_step'],2), [(k['kernel'][4:], round(k['avg_ms'],3)) for k in d['kernels']])"
done
