# exp as exp2 with the folded constant, asm unpack of packed addresses, dense rows within reach of one address
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "dense or tang or config2 or fly or weighted or mixed_degree or full_size or gradient" > gpurun_out/s18_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s18_pytest.log
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'))"; }
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-api --steps 50 --cpu-seconds 2 > gpurun_out/s18_$name.json 2> gpurun_out/s18_$name.err || tail -3 gpurun_out/s18_$name.err | cut -c1-200; echo -n "$name: "; show gpurun_out/s18_$name.json; }
run tang --config tang2019 --dtype f32 --no-f32
run tanggrad --config tang2019 --dtype f32 --gradient --no-f32
run c2 --config 2 --dtype f32 --no-f32
run c2f64 --config 2 --dtype f64 --no-f32
run f64 --dtype f64 --no-f32
run f32 --dtype f32 --no-f32
