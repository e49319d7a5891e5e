# the dense product in double on the current loop (was 1.81 against 2.26 M pairs/s before the loop was restructured)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'))"; }
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-api --steps 20 --cpu-seconds 2 --no-f32 --no-cpu-baseline > gpurun_out/s33_$name.json 2> gpurun_out/s33_$name.err || tail -3 gpurun_out/s33_$name.err | cut -c1-300; echo -n "$name: "; show gpurun_out/s33_$name.json; }
run tang64 --config tang2019 --dtype f64
GD_HIPCC_EXTRA=-DGD_FLY_DENSE=2 run tang64_dense --config tang2019 --dtype f64
