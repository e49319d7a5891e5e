#!/bin/bash
# A/B of compiler flag sets (GD_HIPCC_EXTRA) over bench lines; flag sets are
# separated by '|':  bash scripts/ab_flags.sh "|-DX=1|-DX=1 -DY=2" ["tag:args" ...]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
IFS='|' read -ra sets <<< "$1"; shift
tags=("$@")
[ ${#tags[@]} -eq 0 ] && tags=("f32:--dtype f32" "f64:")
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1], round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
for rep in 1 2; do
i=0
for fl in "${sets[@]}"; do
  for tag in "${tags[@]}"; do
    name=${tag%%:*}; args=${tag#*:}
    GD_HIPCC_EXTRA="$fl" timeout 900 python bench.py --no-api --no-cpu-baseline $args > gpurun_out/abf_${i}_$name.json 2> gpurun_out/abf_${i}_$name.err || tail -5 gpurun_out/abf_${i}_$name.err
    echo -n "[$fl] $name: "; show gpurun_out/abf_${i}_$name.json
  done
  i=$((i+1))
done
done
