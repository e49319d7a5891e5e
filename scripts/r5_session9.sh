#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "streamed or large_pair" > gpurun_out/s9_pytest_large.log 2>&1
log "pytest large rc=$?"; tail -3 gpurun_out/s9_pytest_large.log | cut -c1-300
for ng in 1 4 16; do
  timeout 400 python bench.py --config large --graphs $ng --dtype f64 --steps 5 --warmup 2 --no-api --no-f32 --cpu-seconds 3 > gpurun_out/s9_large${ng}_f64.json 2> gpurun_out/s9_large${ng}_f64.err
  log "large $ng graphs f64 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s9_large${ng}_f64.json').read().strip().splitlines()[-1])
print(d['config']['pairs'], 'pairs', round(d['ms_per_step'],3), 'ms/step', d['accuracy']['max_rel_err_vs_converged_oracle'], [(k['kernel'],k['pairs'],k['grid'],round(k['isolated_ms'],3)) for k in d['kernels']])"
done
timeout 600 python bench.py --config large --steps 5 --warmup 2 --cpu-seconds 4 > gpurun_out/s9_large32_f32.json 2> gpurun_out/s9_large32_f32.err
log "large 32 graphs f32 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s9_large32_f32.json').read().strip().splitlines()[-1])
print(d['value'], round(d['ms_per_step'],3), d['accuracy']['max_rel_err_vs_converged_oracle'], d['other_arithmetic']['ms_per_step'], d['other_arithmetic'].get('max_rel_err_vs_converged_oracle'), d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'], d['api_inclusive'])"
log done
