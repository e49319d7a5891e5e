#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
timeout 1500 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x -k "ranks_through" > gpurun_out/s6_pytest_dist.log 2>&1
log "pytest dist rc=$?"; tail -4 gpurun_out/s6_pytest_dist.log
timeout 600 python scripts/gpr_step_sim.py --world 8 > gpurun_out/s6_gpr_step_sim_f64_lowprio.log 2>&1
log "gpr sim f64 low priority rc=$?"; tail -6 gpurun_out/s6_gpr_step_sim_f64_lowprio.log
GD_DETACHED_PRIORITY=0 timeout 600 python scripts/gpr_step_sim.py --world 8 > gpurun_out/s6_gpr_step_sim_f64_normal.log 2>&1
log "gpr sim f64 normal priority rc=$?"; tail -6 gpurun_out/s6_gpr_step_sim_f64_normal.log
timeout 400 python bench.py --gpr --gpus 2 --share-devices --graphs 300 --steps 3 --warmup 1 > gpurun_out/s6_gpr2.json 2> gpurun_out/s6_gpr2.err
log "gpr 2 ranks rc=$?"; tail -c 700 gpurun_out/s6_gpr2.json; echo
log done
