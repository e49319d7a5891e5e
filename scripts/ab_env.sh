#!/bin/bash
# A/B of an environment switch over the bench lines:
#   bash scripts/ab_env.sh VAR "v1 v2" ["tag:args" ...]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
var=$1; vals=$2; shift 2
tags=("$@")
[ ${#tags[@]} -eq 0 ] && tags=("f32:--dtype f32" "f64:" "g32:--gradient --dtype f32" "g64:--gradient")
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('cpu_baseline') or {}; print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', c.get('max_rel_diff_vs_gpu'), c.get('gradient_max_violation_of_elementwise_bound'), [(k['kernel'].split('_oc')[-1], round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
for rep in 1 2; do
for v in $vals; do
  for tag in "${tags[@]}"; do
    name=${tag%%:*}; args=${tag#*:}
    extra="--no-cpu-baseline"; [ $rep = 1 ] && extra=""
    env $var=$v timeout 900 python bench.py --no-api $extra $args > gpurun_out/ab_${v}_$name.json 2> gpurun_out/ab_${v}_$name.err || tail -5 gpurun_out/ab_${v}_$name.err
    echo -n "$var=$v $name: "; show gpurun_out/ab_${v}_$name.json
  done
done
done
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
