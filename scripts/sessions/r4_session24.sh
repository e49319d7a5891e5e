# packed two-term evaluation of the edge microkernel in the dense product
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -x -k "dense or tang or fly or mixed_degree or maximin" > gpurun_out/s24_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s24_pytest.log
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'))"; }
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-api --steps 50 --cpu-seconds 2 > gpurun_out/s24_$name.json 2> gpurun_out/s24_$name.err || tail -3 gpurun_out/s24_$name.err | cut -c1-200; echo -n "$name: "; show gpurun_out/s24_$name.json; }
run tang --config tang2019 --dtype f32 --no-f32
run tanggrad --config tang2019 --dtype f32 --gradient --no-f32
export GD_PACKED_EDGES=0
run tang_off --config tang2019 --dtype f32 --no-f32
run tanggrad_off --config tang2019 --dtype f32 --gradient --no-f32
