#!/bin/bash
# round 3, session 14: first API call with staged transfers and the native job list
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python scripts/first_api_call.py --f64 2>&1 | tail -12
python scripts/first_api_call.py --f64 --profile 2>&1 | tail -30
python scripts/first_api_call.py 2>&1 | tail -2
for b in 1 16 128; do python scripts/profile_small_call.py $b 2>&1 | head -1; done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_f64_api.json 2> gpurun_out/bench_f64_api.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_f64_api.json').read().strip().split('\n')[-1]); print(d['value'], d['api_inclusive'])"
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
