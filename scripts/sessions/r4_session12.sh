cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "dense or on_the_fly or maximin or molecular" 2>&1 | tail -3
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', d.get('mean_cg_iterations'), [(k['kernel'].split('_oc')[-1], k['pairs'], round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('cpu_baseline') or {}).get('max_rel_diff_vs_gpu'))"; }
run() { name=$1; shift; env "$@" > gpurun_out/s12_$name.json 2> gpurun_out/s12_$name.err || tail -3 gpurun_out/s12_$name.err; echo -n "$name: "; show gpurun_out/s12_$name.json; }
run tang_f32 timeout 600 python bench.py --config tang2019 --dtype f32 --no-api --no-f32 --steps 30 --cpu-seconds 3
run tang_f32_all GD_HIPCC_EXTRA=-DGD_FLY_DENSE=2 timeout 600 python bench.py --config tang2019 --dtype f32 --no-api --no-cpu-baseline --no-f32 --steps 30
run tanggrad_f32_all GD_HIPCC_EXTRA=-DGD_FLY_DENSE=2 timeout 600 python bench.py --config tang2019 --dtype f32 --gradient --no-api --no-cpu-baseline --no-f32 --steps 30
run tang_f64_all GD_HIPCC_EXTRA=-DGD_FLY_DENSE=2 timeout 600 python bench.py --config tang2019 --dtype f64 --no-api --no-cpu-baseline --no-f32 --steps 30
BENCH_ARGS="--config tang2019 --dtype f32" bash scripts/pmc_quick.sh 2>&1 | tail -8
python scripts/profile_first_call.py f64 2>&1 | head -5
