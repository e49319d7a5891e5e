#!/bin/bash
# round 3, session 17: value + gradient on the fly (dense graphs)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "on_the_fly or mixed_degree" > gpurun_out/pytest_fly.log 2>&1; tail -15 gpurun_out/pytest_fly.log
for dt in f32 f64; do
timeout 900 python bench.py --config tang2019 --gradient --dtype $dt --no-cpu-baseline --no-api > gpurun_out/bench_tanggrad_$dt.json 2> gpurun_out/bench_tanggrad_$dt.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_tanggrad_$dt.json').read().strip().split('\n')[-1])
print('$dt', round(d['value']/1e6,3), 'M pairs/s', round(d['ms_per_step'],2), 'ms')
for k in d['kernels']: print('   %-40s %6d pairs iso %.3f ms iters %.1f'%(k['kernel'],k['pairs'],k['isolated_ms'],k['mean_cg_iterations']))
PY
done
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
