#!/bin/bash
# Launch-merging threshold against step time for shard-sized job lists
# (354 graphs = 1/8, 500 graphs = 1/4 of the 1000-graph matrix).
set -u
cd "$GRAFT_REPO_ROOT"
for g in 354 500 708; do
for a in "--dtype f32" ""; do
for m in 2048 6144 8192 12288 16384; do
  r=""
  for rep in 1 2 3; do
    v=$(GD_MIN_LAUNCH=$m python bench.py --graphs $g --steps 200 --no-cpu-baseline --no-api --isolated-steps 0 $a | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), len(d['kernels']))")
    r="$r | $v"
  done
  echo "graphs=$g ${a:-f64} min_launch=$m $r"
done; done; done
