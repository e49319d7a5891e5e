#!/bin/bash
# round 3, session 18: configuration 2 in double -- the 16-wave slot variants against the on-the-fly ones
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print(sys.argv[1], round(d['ms_per_step'],3),'ms')
for k in d['kernels']: print('   %-34s %6d pairs  avg %.3f iso %.3f'%(k['kernel'],k['pairs'],k['avg_ms'],k['isolated_ms']))
PY
}
A="--config 2 --no-cpu-baseline --no-api"
python bench.py $A --dtype f64 > gpurun_out/c2x_base64.json 2>/dev/null; show gpurun_out/c2x_base64.json
M="1:32:3:8,1:48:5:8,1:64:9:8,4:32:3:8,4:40:2:8,4:48:4:8,4:64:5:8,8:40:2:8,8:48:3:8,8:64:4:8,4:0:1:0,4:0:2:0,4:0:3:0,8:0:2:0,16:0:2:0,16:0:4:0,16:64:8"
GD_NATIVE_HOST=0 GD_FLY_MIN_DEGREE=0 GD_VARIANTS="$M" python bench.py $A --dtype f64 > gpurun_out/c2x_fly64.json 2>gpurun_out/c2x_fly64.err; show gpurun_out/c2x_fly64.json
GD_NATIVE_HOST=0 GD_FLY_MIN_DEGREE=0 GD_VARIANTS="$M" python bench.py $A --dtype f32 > gpurun_out/c2x_fly32.json 2>gpurun_out/c2x_fly32.err; show gpurun_out/c2x_fly32.json
