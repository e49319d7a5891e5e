#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python scripts/oc_sweep.py --config2 --waves=1,2,3,4 > gpurun_out/sweep_c2.log 2>&1
tail -14 gpurun_out/sweep_c2.log
GD_HIPCC_EXTRA="-DGD_OC_PACK=0" timeout 900 python bench.py --config 2 --no-cpu-baseline --no-api --no-f32 > gpurun_out/bench_c2_nopack.json 2> gpurun_out/bench_c2_nopack.err
head -c 200 gpurun_out/bench_c2_nopack.json; echo
for tag in "c2:--config 2" "c2f64:--config 2 --dtype f64" "nws48:--config nws48" "f64:"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "bench $name rc=$?"; tail -c 200 gpurun_out/bench_$name.err; head -c 200 gpurun_out/bench_$name.json; echo
done
timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or multi_wave or nodelabeled or weighted" > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
