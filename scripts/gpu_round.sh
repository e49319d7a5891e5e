#!/bin/bash
# One gpurun session: GPU tests and the bench lines.
# Usage (from the repo root, through gpurun):  bash scripts/gpu_round.sh [tests|bench|all|scale]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
what=${1:-all}
if [ "$what" = tests ] || [ "$what" = all ]; then
  timeout 2400 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -30 gpurun_out/pytest_gpu.log
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
  for tag in "f64:" "f32:--dtype f32" "grad64:--gradient" "grad32:--gradient --dtype f32" "c2:--config 2" "c2f64:--config 2 --dtype f64" "gpr:--gpr" "gpr64:--gpr --dtype f64"; do
    name=${tag%%:*}; args=${tag#*:}
    timeout 900 python bench.py $args > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
    echo "bench $name rc=$?"; head -c 300 gpurun_out/bench_$name.json; echo
  done
fi
if [ "$what" = scale ] || [ "$what" = all ]; then
  # the multi-rank launch line of the driver, two ranks sharing this GPU
  # (gloo collective on host memory: correctness of the N > 1 path)
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 5 --warmup 2 --share-devices > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
  echo "bench 2 ranks rc=$?"; tail -c 700 gpurun_out/bench_2ranks.json; echo
fi
# (the JIT cache is pre-built by __graft_entry__.build() and travels with the
# snapshot: nothing to bring back -- thousands of code objects would exceed
# what gpurun merges into gpurun_out/)
