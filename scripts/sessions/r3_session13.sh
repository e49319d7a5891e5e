#!/bin/bash
# round 3, session 13: host path -- module-set memo, composite class cache, pinned staging of large uploads
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 600 python scripts/first_call.py --f64 > gpurun_out/first_call_f64.log 2>&1; grep "trial\|repeat" gpurun_out/first_call_f64.log
timeout 600 python scripts/first_call.py > gpurun_out/first_call_f32.log 2>&1; grep "trial\|repeat" gpurun_out/first_call_f32.log
for b in 1 16 128; do python scripts/profile_small_call.py $b 2>&1 | head -1; done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_f64_api.json 2> gpurun_out/bench_f64_api.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_f64_api.json').read().strip().split('\n')[-1]); print(d['value'], d['api_inclusive'])"
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
