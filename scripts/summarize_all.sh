#!/bin/bash
# profiles/<round>_<name>_{kernel_stats.csv,pmc.csv,bench.json} + traffic.json
# from the gpurun_out/prof_<name> directories of scripts/profile_all.sh
set -eu
round=${1:-r06}
for name in f64 f32 grad64 grad32 c2 c2f64 tang tanggrad large; do
  [ -d gpurun_out/prof_$name ] || continue
  rm -rf gpurun_out/prof && cp -r gpurun_out/prof_$name gpurun_out/prof
  python scripts/summarize_profile.py ${round}_$name $name > /dev/null
  echo "$name done"
done
rm -rf gpurun_out/prof
