#!/bin/bash
# Round 5, session 5: multi-rank tests, the N-rank GPR step simulation,
# configuration 2's occupancy sweep (the alternatives to the spilling forms).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
timeout 1500 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x -k "ranks_through" > gpurun_out/s5_pytest_dist.log 2>&1
log "pytest dist rc=$?"; tail -4 gpurun_out/s5_pytest_dist.log
timeout 600 python scripts/gpr_step_sim.py > gpurun_out/s5_gpr_step_sim_f64.log 2>&1
log "gpr sim f64 rc=$?"; tail -12 gpurun_out/s5_gpr_step_sim_f64.log
timeout 600 python scripts/gpr_step_sim.py --f32 > gpurun_out/s5_gpr_step_sim_f32.log 2>&1
log "gpr sim f32 rc=$?"; tail -12 gpurun_out/s5_gpr_step_sim_f32.log
timeout 600 python scripts/oc_sweep.py --config2 > gpurun_out/s5_c2_oc_sweep_f32.log 2>&1
log "sweep f32 rc=$?"; cat gpurun_out/s5_c2_oc_sweep_f32.log
timeout 600 python scripts/oc_sweep.py --config2 --f64 > gpurun_out/s5_c2_oc_sweep_f64.log 2>&1
log "sweep f64 rc=$?"; cat gpurun_out/s5_c2_oc_sweep_f64.log
timeout 400 python bench.py --gpr --gpus 2 --share-devices --graphs 300 --steps 3 --warmup 1 > gpurun_out/s5_gpr2.json 2> gpurun_out/s5_gpr2.err
log "gpr 2 ranks rc=$?"; tail -c 900 gpurun_out/s5_gpr2.json; echo
timeout 400 python bench.py --gpus 2 --share-devices --graphs 300 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s5_sharded2.json 2> gpurun_out/s5_sharded2.err
log "sharded 2 ranks rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s5_sharded2.json').read().strip().splitlines()[-1]); print(d['phases_per_rank'], d['sharded_check'])"
timeout 300 python bench.py --sharded --steps 50 --no-cpu-baseline > gpurun_out/s5_sharded1.json 2> gpurun_out/s5_sharded1.err
log "sharded 1 rank (nccl) rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/s5_sharded1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phases_per_rank'])"
log done
