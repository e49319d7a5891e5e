#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -k "dense or config2 or multi_wave or gradient" > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
for w in 3 4 5 6; do
  GD_FLY_WAVES=$w timeout 600 python bench.py --config tang2019 --no-cpu-baseline --no-api --no-f32 > gpurun_out/bench_tang_w$w.json 2> gpurun_out/bench_tang_w$w.err
  echo "tang waves=$w: $(head -c 130 gpurun_out/bench_tang_w$w.json | cut -c40-130)"
done
for u in 2 8; do
  GD_HIPCC_EXTRA="-DGD_FLY_U=$u" timeout 600 python bench.py --config tang2019 --no-cpu-baseline --no-api --no-f32 > gpurun_out/bench_tang_u$u.json 2> gpurun_out/bench_tang_u$u.err
  echo "tang U=$u: $(head -c 130 gpurun_out/bench_tang_u$u.json | cut -c40-130)"
done
timeout 600 python bench.py --config tang2019 --dtype f64 --no-cpu-baseline --no-api --no-f32 > gpurun_out/bench_tang64.json 2> gpurun_out/bench_tang64.err
echo "tang f64: $(head -c 130 gpurun_out/bench_tang64.json | cut -c40-130)"
for tag in "grad64:--gradient" "c2:--config 2"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args --no-cpu-baseline --no-api > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "bench $name: $(head -c 130 gpurun_out/bench_$name.json | cut -c40-130)"
done
