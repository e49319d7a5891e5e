#!/bin/bash
# round 3, session 15: api_inclusive of bench.py, three runs per arithmetic
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for dt in f64 f32; do for k in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --dtype $dt > gpurun_out/bench_api.json 2> gpurun_out/bench_api.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_api.json').read().strip().split('\n')[-1]); a=d['api_inclusive']; print('$dt', round(d['value']/1e6,1), round(a['first_call_ms'],1), a['repeat_calls_ms'])"
done; done
python scripts/first_api_call.py --f64 2>&1 | tail -12
