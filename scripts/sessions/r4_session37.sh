# compute side of the strong scaling of configurations 4 / 5 with the round-4 kernels: every rank's shard stepped on one GPU
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python scripts/shard_sim.py > gpurun_out/r4_shard_sim_f64.log 2>&1; tail -8 gpurun_out/r4_shard_sim_f64.log
timeout 1200 python scripts/shard_sim.py --f32 > gpurun_out/r4_shard_sim_f32.log 2>&1; tail -8 gpurun_out/r4_shard_sim_f32.log
timeout 1200 python scripts/shard_sim.py --gradient > gpurun_out/r4_shard_sim_grad64.log 2>&1; tail -8 gpurun_out/r4_shard_sim_grad64.log
