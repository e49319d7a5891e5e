#!/bin/bash
# round 3, GPU session 3: tests, first-call breakdown, cost calibration,
# shard simulation with the measured plan, bench lines
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -14 gpurun_out/pytest_gpu.log
timeout 300 python scripts/first_call.py --f64 --profile > gpurun_out/first_call_f64.log 2>&1
timeout 300 python scripts/first_call.py > gpurun_out/first_call_f32.log 2>&1
grep -h "trial\|repeat" gpurun_out/first_call_f64.log gpurun_out/first_call_f32.log
timeout 900 python scripts/calibrate_cost.py --out=gpurun_out/cost_table.json > gpurun_out/calibrate.log 2>&1
tail -45 gpurun_out/calibrate.log
for fl in "" "--f32"; do
  GD_COST_TABLE=gpurun_out/cost_table.json timeout 900 python scripts/shard_sim.py $fl --no-merge > gpurun_out/shard_sim_nomerge$fl.log 2>&1
  tail -20 gpurun_out/shard_sim_nomerge$fl.log
done
GD_COST_TABLE=gpurun_out/cost_table.json timeout 900 python scripts/shard_sim.py --gradient --f32 --no-merge --mode=measured > gpurun_out/shard_sim_grad32.log 2>&1
tail -8 gpurun_out/shard_sim_grad32.log
for tag in "f64:" "f32:--dtype f32" "grad64:--gradient" "grad32:--gradient --dtype f32" "gpr:--gpr" "gpr64:--gpr --dtype f64"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "bench $name rc=$?"; head -c 200 gpurun_out/bench_$name.json; echo
done
GD_COST_TABLE=gpurun_out/cost_table.json timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
echo "bench 2 ranks rc=$?"; tail -c 400 gpurun_out/bench_2ranks.json; echo
GD_COST_TABLE=gpurun_out/cost_table.json timeout 900 python bench.py --gpus 2 --gpr --steps 10 --warmup 2 > gpurun_out/bench_gpr_2ranks.json 2> gpurun_out/bench_gpr_2ranks.err
echo "bench gpr 2 ranks rc=$?"; tail -c 600 gpurun_out/bench_gpr_2ranks.json; echo
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
