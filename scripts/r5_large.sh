#!/bin/bash
# Round 5: the large-graph regime (bench.py --config large): streamed solver
# against the round-4 general solver (GD_STREAM=0) on the same pairs.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "streamed or large_pair or undefined_on_empty" > gpurun_out/pytest_large.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_large.log
for dt in f32 f64; do
  GD_STREAM=0 timeout 900 python bench.py --config large --graphs 16 --dtype $dt --steps 2 --warmup 1 --no-api --no-f32 > gpurun_out/bench_large16_general_$dt.json 2> gpurun_out/bench_large16_general_$dt.err
  echo "general $dt rc=$?"; head -c 400 gpurun_out/bench_large16_general_$dt.json; echo
  timeout 900 python bench.py --config large --graphs 16 --dtype $dt --steps 3 --warmup 1 --no-api --no-f32 > gpurun_out/bench_large16_stream_$dt.json 2> gpurun_out/bench_large16_stream_$dt.err
  echo "stream $dt rc=$?"; head -c 400 gpurun_out/bench_large16_stream_$dt.json; echo
done
timeout 900 python bench.py --config large --dtype f32 > gpurun_out/bench_large_f32.json 2> gpurun_out/bench_large_f32.err
echo "large f32 rc=$?"; head -c 400 gpurun_out/bench_large_f32.json; echo
