#!/bin/bash
# round 3, session 19: full GPU suite, api-inclusive numbers
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=3 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for dt in f64 f32; do for k in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --dtype $dt > gpurun_out/bench_api.json 2> gpurun_out/bench_api.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_api.json').read().strip().split('\n')[-1]); a=d['api_inclusive']; print('$dt', round(d['value']/1e6,1), round(a['first_call_ms'],1), a['repeat_calls_ms'])"
done; done
GD_API_TIMING=1 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep " ms on "
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
