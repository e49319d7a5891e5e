#!/bin/bash
# Round 5, session 7: the streamed solver with several workgroups per pair.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "streamed or large_pair" > gpurun_out/s7_pytest_large.log 2>&1
log "pytest large rc=$?"; tail -5 gpurun_out/s7_pytest_large.log
for ng in 1 2 4 8 16; do
  timeout 300 python bench.py --config large --graphs $ng --steps 5 --warmup 2 --no-api --no-f32 --cpu-seconds 3 > gpurun_out/s7_large${ng}_f32.json 2> gpurun_out/s7_large${ng}_f32.err
  log "large $ng graphs rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s7_large${ng}_f32.json').read().strip().splitlines()[-1])
print(d['config']['pairs'], 'pairs', round(d['ms_per_step'],3), 'ms/step', d['accuracy']['max_rel_err_vs_converged_oracle'], [(k['kernel'],k['pairs'],k['grid'],round(k['isolated_ms'],3)) for k in d['kernels']])"
  GD_STREAM_PARTS=1 timeout 300 python bench.py --config large --graphs $ng --steps 3 --warmup 1 --no-api --no-f32 --no-cpu-baseline > gpurun_out/s7_large${ng}_f32_onepart.json 2> gpurun_out/s7_large${ng}_f32_onepart.err
  log "  one workgroup per pair rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s7_large${ng}_f32_onepart.json').read().strip().splitlines()[-1])
print('  ', round(d['ms_per_step'],3), 'ms/step')"
done
timeout 300 python bench.py --config large --graphs 4 --dtype f64 --steps 5 --warmup 2 --no-api --no-f32 --cpu-seconds 3 > gpurun_out/s7_large4_f64.json 2> gpurun_out/s7_large4_f64.err
log "large 4 graphs f64 rc=$?"; head -c 300 gpurun_out/s7_large4_f64.json; echo
timeout 1500 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x -k "ranks_through" > gpurun_out/s7_pytest_dist.log 2>&1
log "pytest dist rc=$?"; tail -4 gpurun_out/s7_pytest_dist.log
log done
