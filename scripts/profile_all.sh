#!/bin/bash
# scripts/profile.sh for the four profiled workloads, each into its own
# directory under gpurun_out/ (prof_f64, prof_f32, prof_grad64, prof_c2).
set -u
cd "$GRAFT_REPO_ROOT"
for tag in "f64:" "f32:--dtype f32" "grad64:--gradient" "c2:--config 2"; do
  name=${tag%%:*}; args=${tag#*:}
  BENCH_ARGS="$args" bash scripts/profile.sh > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name && mv gpurun_out/prof gpurun_out/prof_$name
  # keep what the summariser reads; drop the bulky per-dispatch traces
  find gpurun_out/prof_$name -name "*kernel_trace.csv" -delete
  echo "$name: $(tail -c 200 gpurun_out/prof_$name/bench.json | head -c 120)"; du -sh gpurun_out/prof_$name
done
