#!/bin/bash
# round 3 GPU session: parity tests, occupancy sweeps of the static-layout
# variants (value and value + gradient, both arithmetics), bench lines
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 600 python scripts/oc_sweep.py --f64 --grad --waves=1,2 > gpurun_out/sweep_f64_grad.log 2>&1
timeout 600 python scripts/oc_sweep.py --grad --waves=2,3,4 > gpurun_out/sweep_f32_grad.log 2>&1
timeout 600 python scripts/oc_sweep.py --f64 --waves=2,3,4 > gpurun_out/sweep_f64.log 2>&1
timeout 600 python scripts/oc_sweep.py --waves=3,4,5,6 > gpurun_out/sweep_f32.log 2>&1
GD_OC_STATIC=0 timeout 600 python scripts/oc_sweep.py --f64 --grad --waves=2 > gpurun_out/sweep_f64_grad_dyn.log 2>&1
GD_OC_STATIC=0 timeout 600 python scripts/oc_sweep.py --grad --waves=2,3 > gpurun_out/sweep_f32_grad_dyn.log 2>&1
tail -n 14 gpurun_out/sweep_*.log
for tag in "f64:" "f32:--dtype f32" "grad64:--gradient" "grad32:--gradient --dtype f32"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args --no-cpu-baseline > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "bench $name rc=$?"; head -c 250 gpurun_out/bench_$name.json; echo
done
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
ls gpurun_out/jit | wc -l
