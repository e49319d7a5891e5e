#!/bin/bash
# round 3: the measurement session behind profiles/r03_* and DESIGN.md
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for tag in "f64:" "f32:--dtype f32" "grad64:--gradient" "grad32:--gradient --dtype f32" "c2:--config 2" "c2f64:--config 2 --dtype f64" "gpr:--gpr" "gpr64:--gpr --dtype f64" "nws48:--config nws48" "tang:--config tang2019"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "bench $name rc=$?"; head -c 160 gpurun_out/bench_$name.json; echo
done
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
echo "bench 2 ranks rc=$?"
timeout 900 python bench.py --gpus 2 --gpr --steps 10 --warmup 2 > gpurun_out/bench_gpr_2ranks.json 2> gpurun_out/bench_gpr_2ranks.err
echo "bench gpr 2 ranks rc=$?"
for fl in "" "--f32" "--gradient --f32" "--gradient"; do
  tag=$(echo "$fl" | tr -d ' -')
  timeout 900 python scripts/shard_sim.py $fl --mode=measured > gpurun_out/shard_sim_$tag.log 2>&1
  grep "full step\|world\|round 2" gpurun_out/shard_sim_$tag.log
done
timeout 300 python scripts/first_call.py --f64 > gpurun_out/first_call_f64.log 2>&1
timeout 300 python scripts/time_cholesky.py > gpurun_out/time_cholesky.log 2>&1
tail -6 gpurun_out/time_cholesky.log
bash scripts/profile_all.sh
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
