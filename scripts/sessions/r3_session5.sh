#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp gpurun_out_cost_table.json gpurun_out/cost_table.json 2>/dev/null
for fl in "" "--f32"; do
  tag=$(echo "$fl" | tr -d ' -')
  timeout 900 python scripts/shard_sim.py $fl --mode=measured > gpurun_out/shard_sim_$tag.log 2>&1
  grep "full step\|world\|rebalance" gpurun_out/shard_sim_$tag.log
done
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err
echo "bench f64 rc=$?"; head -c 200 gpurun_out/bench_f64.json; echo
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
echo "bench 2 ranks rc=$?"; tail -c 300 gpurun_out/bench_2ranks.json; echo
