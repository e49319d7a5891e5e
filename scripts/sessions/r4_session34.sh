# the 18-instruction double exponential: parity of the double builds, configuration 2 and the dense set in double
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x -k "not float32" > gpurun_out/s34_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s34_pytest.log
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'))"; }
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-api --steps 30 --cpu-seconds 2 --no-f32 > gpurun_out/s34_$name.json 2> gpurun_out/s34_$name.err || tail -3 gpurun_out/s34_$name.err | cut -c1-300; echo -n "$name: "; show gpurun_out/s34_$name.json; }
run c2f64 --config 2 --dtype f64
run tang64 --config tang2019 --dtype f64
run f64 --dtype f64
run grad64 --dtype f64 --gradient
GD_HIPCC_EXTRA=-DGD_EXP_OCML run c2f64_ocml --config 2 --dtype f64
