#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
for t in "large_pair" "streamed_solver_on_large_spatial_graphs and float32" "streamed_solver_on_large_spatial_graphs and float64" "several_workgroups and float32" "several_workgroups and float64"; do
  name=$(echo "$t" | tr ' ' '_')
  timeout 240 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "$t" > gpurun_out/s8_$name.log 2>&1
  log "pytest [$t] rc=$?"; tail -3 gpurun_out/s8_$name.log | cut -c1-300
done
log done
