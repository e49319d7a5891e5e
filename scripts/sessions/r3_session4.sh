#!/bin/bash
# round 3, GPU session 4: calibration with per-variant tails, shard
# simulation with the global merge map, the new bench configurations
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
timeout 900 python scripts/calibrate_cost.py --out=gpurun_out/cost_table.json > gpurun_out/calibrate.log 2>&1
tail -3 gpurun_out/calibrate.log
for fl in "" "--f32" "--gradient --f32" "--gradient"; do
  tag=$(echo "$fl" | tr -d ' -')
  GD_COST_TABLE=gpurun_out/cost_table.json timeout 900 python scripts/shard_sim.py $fl --mode=measured > gpurun_out/shard_sim_$tag.log 2>&1
  grep "full step\|world" gpurun_out/shard_sim_$tag.log
done
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for tag in "nws48:--config nws48" "tang:--config tang2019" "tang64:--config tang2019 --dtype f64" "f64:"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "bench $name rc=$?"; tail -c 300 gpurun_out/bench_$name.err; head -c 300 gpurun_out/bench_$name.json; echo
done
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
