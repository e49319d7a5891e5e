#!/bin/bash
# Round 5, session 4: multi-rank tests, remaining fuzz modes, the 8-rank GPR
# step simulation, profile of bench.py --config large.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
timeout 1500 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x -k "ranks_through" > gpurun_out/s4_pytest_dist.log 2>&1
log "pytest dist rc=$?"; tail -4 gpurun_out/s4_pytest_dist.log
timeout 1200 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -k "spatial or sharded" > gpurun_out/s4_pytest_fuzz.log 2>&1
log "pytest fuzz rc=$?"; tail -6 gpurun_out/s4_pytest_fuzz.log
timeout 600 python scripts/gpr_step_sim.py > gpurun_out/s4_gpr_step_sim_f64.log 2>&1
log "gpr sim f64 rc=$?"; cat gpurun_out/s4_gpr_step_sim_f64.log | tail -12
timeout 600 python scripts/gpr_step_sim.py --f32 > gpurun_out/s4_gpr_step_sim_f32.log 2>&1
log "gpr sim f32 rc=$?"; cat gpurun_out/s4_gpr_step_sim_f32.log | tail -12
BENCH_ARGS="--config large --cpu-seconds 4" timeout 1500 bash scripts/profile.sh > /dev/null 2>&1
log "profile large rc=$?"; ls gpurun_out/prof; head -c 400 gpurun_out/prof/bench.json; echo
rm -rf gpurun_out/prof_large && mv gpurun_out/prof gpurun_out/prof_large
find gpurun_out/prof_large -name "*kernel_trace.csv" -delete
log done
