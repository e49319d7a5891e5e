#!/bin/bash
# round 3, session 16: tests and the API-side numbers after the host-path work
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for tag in "f64:" "f32:--dtype f32" "grad64:--gradient" "grad32:--gradient --dtype f32" "c2:--config 2" "c2f64:--config 2 --dtype f64" "gpr:--gpr" "gpr64:--gpr --dtype f64" "nws48:--config nws48" "tang:--config tang2019"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "bench $name rc=$?"; head -c 150 gpurun_out/bench_$name.json; echo
done
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
echo "bench 2 ranks rc=$? lines $(wc -l < gpurun_out/bench_2ranks.json)"
timeout 900 python bench.py --gpus 2 --gpr --steps 10 --warmup 2 > gpurun_out/bench_gpr_2ranks.json 2> gpurun_out/bench_gpr_2ranks.err
echo "bench gpr 2 ranks rc=$? lines $(wc -l < gpurun_out/bench_gpr_2ranks.json)"
timeout 300 python scripts/first_call.py --f64 > gpurun_out/first_call_f64.log 2>&1; grep "trial\|repeat" gpurun_out/first_call_f64.log
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
