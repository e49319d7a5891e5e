#!/bin/bash
# scripts/profile.sh for every profiled workload, each into its own directory
# under gpurun_out/ (prof_<name>); condensed afterwards, in the build
# container, by  scripts/summarize_all.sh r06  into profiles/ (every workload
# of profiles/traffic.json at one HEAD: round 6).
set -u
cd "$GRAFT_REPO_ROOT"
for tag in ${PROFILE_TAGS:-"f64:" "f32:--dtype f32" "grad64:--gradient" "grad32:--gradient --dtype f32" "c2:--config 2" "c2f64:--config 2 --dtype f64" "tang:--config tang2019" "tanggrad:--config tang2019 --gradient" "large:--config large"}; do
  name=${tag%%:*}; args=${tag#*:}
  BENCH_ARGS="$args" bash scripts/profile.sh > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name && mv gpurun_out/prof gpurun_out/prof_$name
  # keep what the summariser reads; drop the bulky per-dispatch traces
  find gpurun_out/prof_$name -name "*kernel_trace.csv" -delete
  echo "$name: $(tail -c 200 gpurun_out/prof_$name/bench.json | head -c 120)"; du -sh gpurun_out/prof_$name
done
