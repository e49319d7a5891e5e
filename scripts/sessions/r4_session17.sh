cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "
from graphdot_amd.hip import runtime
p = runtime.device_props(); print('lds_per_block', p.lds_per_block, 'CUs', p.compute_units)"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'))"; }
run() { name=$1; shift; env "$@" timeout 600 python bench.py --config 2 --dtype f64 --no-api --no-f32 --steps 30 --cpu-seconds 2 > gpurun_out/s17_$name.json 2> gpurun_out/s17_$name.err || tail -3 gpurun_out/s17_$name.err | cut -c1-200; echo -n "$name: "; show gpurun_out/s17_$name.json; }
run sl10 GD_HIPCC_EXTRA=-DGD_OC_SL=10
run sl8 GD_HIPCC_EXTRA=-DGD_OC_SL=8
run sl6 GD_HIPCC_EXTRA=-DGD_OC_SL=6
run sl4 GD_HIPCC_EXTRA=-DGD_OC_SL=4
