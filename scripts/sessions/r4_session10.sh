cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "dense or on_the_fly or maximin or molecular" > gpurun_out/s10_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s10_pytest.log
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', d.get('mean_cg_iterations'), [(k['kernel'].split('_oc')[-1], k['pairs'], round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
run() { name=$1; shift; env "$@" > gpurun_out/s10_$name.json 2> gpurun_out/s10_$name.err || tail -3 gpurun_out/s10_$name.err; echo -n "$name: "; show gpurun_out/s10_$name.json; }
for dt in f32 f64; do
run tang_${dt}_dense timeout 600 python bench.py --config tang2019 --dtype $dt --no-api --no-cpu-baseline --no-f32 --steps 30
run tang_${dt}_csr GD_HIPCC_EXTRA=-DGD_FLY_DENSE=0 timeout 600 python bench.py --config tang2019 --dtype $dt --no-api --no-cpu-baseline --no-f32 --steps 30
run tanggrad_${dt}_dense timeout 600 python bench.py --config tang2019 --dtype $dt --gradient --no-api --no-cpu-baseline --no-f32 --steps 30
run tanggrad_${dt}_csr GD_HIPCC_EXTRA=-DGD_FLY_DENSE=0 timeout 600 python bench.py --config tang2019 --dtype $dt --gradient --no-api --no-cpu-baseline --no-f32 --steps 30
done
