cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
run() { name=$1; shift; env "$@" timeout 600 python bench.py --config 2 --dtype f64 --no-api --no-cpu-baseline --no-f32 --steps 30 > gpurun_out/s8_$name.json 2> gpurun_out/s8_$name.err || tail -3 gpurun_out/s8_$name.err; echo -n "$name: "; show gpurun_out/s8_$name.json; }
run packd_on X=1
run packd_off GD_HIPCC_EXTRA=-DGD_OC_PACK=3
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "multi_wave or random_graphs or config2 or mixed_degree" 2>&1 | tail -3
