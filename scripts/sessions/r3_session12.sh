#!/bin/bash
# round 3, session 12: cost table of the sharding planner refitted on the kernels with the grid walk; shard_sim with it
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python scripts/calibrate_cost.py --out=gpurun_out/cost_table.json > gpurun_out/calibrate_cost.log 2>&1
tail -5 gpurun_out/calibrate_cost.log
export GD_COST_TABLE=$PWD/gpurun_out/cost_table.json
for fl in "" "--f32" "--gradient --f32" "--gradient"; do
  tag=$(echo "$fl" | tr -d ' -')
  timeout 900 python scripts/shard_sim.py $fl --mode=measured > gpurun_out/shard_sim_$tag.log 2>&1
  grep "full step\|world" gpurun_out/shard_sim_$tag.log
done
timeout 600 python -m pytest tests/test_distributed_gpu.py -q -x 2>&1 | tail -3
