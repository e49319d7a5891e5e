#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
for t in "dense_tile and True" "dense_tile and False" "large_pair" "several_workgroups and float32" "several_workgroups and float64"; do
  name=$(echo "$t" | tr ' ' '_')
  timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "$t" > gpurun_out/s12_$name.log 2>&1
  log "pytest [$t] rc=$?"; grep -n "^E  " gpurun_out/s12_$name.log | head -12 | cut -c1-400; tail -2 gpurun_out/s12_$name.log | cut -c1-300
done
timeout 400 python bench.py --config large --graphs 8 --gradient --steps 3 --warmup 1 --no-api --cpu-seconds 3 > gpurun_out/s12_large8_grad_f32.json 2> gpurun_out/s12_large8_grad_f32.err
log "large 8 graphs gradient rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s12_large8_grad_f32.json').read().strip().splitlines()[-1])
print(d['config']['pairs'], 'pairs', round(d['ms_per_step'],3), 'ms/step', d['cpu_baseline'].get('gradient_max_violation_of_elementwise_bound'), [(k['kernel'],k['pairs'],k['grid'],round(k['isolated_ms'],3)) for k in d['kernels']])"; tail -3 gpurun_out/s12_large8_grad_f32.err
GD_STREAM=0 timeout 400 python bench.py --config large --graphs 8 --gradient --steps 2 --warmup 1 --no-api --no-cpu-baseline > gpurun_out/s12_large8_grad_f32_general.json 2> gpurun_out/s12_large8_grad_f32_general.err
log "  general solver rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s12_large8_grad_f32_general.json').read().strip().splitlines()[-1])
print('  ', round(d['ms_per_step'],3), 'ms/step')"
log done
