cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', d.get('mean_cg_iterations'), [(k['kernel'].split('_oc')[-1], k['pairs'], round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('cpu_baseline') or {}).get('max_rel_diff_vs_gpu'))"; }
run() { name=$1; shift; env "$@" > gpurun_out/s13_$name.json 2> gpurun_out/s13_$name.err || tail -3 gpurun_out/s13_$name.err; echo -n "$name: "; show gpurun_out/s13_$name.json; }
run tang_f32 timeout 600 python bench.py --config tang2019 --dtype f32 --no-api --no-f32 --steps 30 --cpu-seconds 3
run tanggrad_f32 timeout 600 python bench.py --config tang2019 --dtype f32 --gradient --no-api --no-cpu-baseline --no-f32 --steps 30
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/s13_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s13_pytest.log
