#!/bin/bash
# round 3, session 10: first-call profile; A/B of the grid walk and the scalar header loads
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python scripts/first_call.py --f64 --profile > gpurun_out/first_call_f64.log 2>&1
tail -45 gpurun_out/first_call_f64.log
for ab in "grid0:-DGD_OC_GRID=0" "sload0:-DGD_OC_SLOAD=0" "both0:-DGD_OC_GRID=0 -DGD_OC_SLOAD=0"; do
  name=${ab%%:*}; fl=${ab#*:}
  for dt in f64 f32; do
    GD_HIPCC_EXTRA="$fl" timeout 900 python bench.py --dtype $dt --no-cpu-baseline --no-api > gpurun_out/ab_${name}_$dt.json 2> gpurun_out/ab_${name}_$dt.err
    echo "ab $name $dt rc=$?"; head -c 200 gpurun_out/ab_${name}_$dt.json; echo
  done
done
