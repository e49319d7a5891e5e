# float scalars (FSCAL) in the double CG + SEQ: full GPU suite, then the bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q -x > gpurun_out/s3_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s3_pytest.log
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', d.get('mean_cg_iterations'), d.get('accuracy',{}) and d['accuracy'].get('max_rel_err_vs_converged_oracle'), (d.get('fp64_converged') or {}).get('value'), (d.get('fp64_converged') or {}).get('max_rel_err_vs_converged_oracle'), [(k['kernel'].split('_oc')[-1].replace('_tab',''), k['pairs'], round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
run() { name=$1; shift; env "$@" > gpurun_out/s3_$name.json 2> gpurun_out/s3_$name.err || tail -3 gpurun_out/s3_$name.err; echo -n "$name: "; show gpurun_out/s3_$name.json; }
run f64 timeout 900 python bench.py --no-api
run f64_nofscal GD_HIPCC_EXTRA=-DGD_OC_FSCAL=0 timeout 900 python bench.py --no-api --no-cpu-baseline --no-f32
run grad64 timeout 900 python bench.py --gradient --no-api --no-cpu-baseline
run c2f64 timeout 900 python bench.py --config 2 --dtype f64 --no-api --no-cpu-baseline --no-f32
run c2f64_nofscal GD_HIPCC_EXTRA=-DGD_OC_FSCAL=0 timeout 900 python bench.py --config 2 --dtype f64 --no-api --no-cpu-baseline --no-f32
run tang64 timeout 900 python bench.py --config tang2019 --dtype f64 --no-api --no-cpu-baseline --no-f32
