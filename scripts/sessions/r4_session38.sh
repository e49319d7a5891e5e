# cost table of the shard planner re-measured on the round-4 kernels, then the shard simulation again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python scripts/calibrate_cost.py --out=gpurun_out/cost_table.json 2>&1 | tail -5
export GD_COST_TABLE=$PWD/gpurun_out/cost_table.json
timeout 1200 python scripts/shard_sim.py > gpurun_out/r4_shard_sim_f64.log 2>&1; grep -E "^world|rebalance" gpurun_out/r4_shard_sim_f64.log | tail -9
timeout 1200 python scripts/shard_sim.py --f32 > gpurun_out/r4_shard_sim_f32.log 2>&1; grep -E "^world 8" gpurun_out/r4_shard_sim_f32.log
timeout 1200 python scripts/shard_sim.py --gradient > gpurun_out/r4_shard_sim_grad64.log 2>&1; grep -E "^world 8" gpurun_out/r4_shard_sim_grad64.log
timeout 1200 python scripts/shard_sim.py --gradient --f32 > gpurun_out/r4_shard_sim_grad32.log 2>&1; grep -E "^world 8" gpurun_out/r4_shard_sim_grad32.log
