# SEQ tuned (occupancy table, combined tolerance): parity subset, grad64 bench, value-kernel occupancy probes, PMC table of grad64 and f64
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_gpr.py -m gpu -q -x -k "gradient or fp64 or gpr or fit" > gpurun_out/s5_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s5_pytest.log
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', d.get('mean_cg_iterations'), [(k['kernel'].split('_oc4_')[-1].replace('_tab','').split('_L')[-1], round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
run() { name=$1; shift; env "$@" > gpurun_out/s5_$name.json 2> gpurun_out/s5_$name.err || tail -3 gpurun_out/s5_$name.err; echo -n "$name: "; show gpurun_out/s5_$name.json; }
run grad64 timeout 900 python bench.py --gradient --no-api --no-cpu-baseline --steps 50
run f64 timeout 900 python bench.py --no-api --no-cpu-baseline --no-f32 --steps 100
run f64_occ3 GD_OCCUPANCY=1:26:3,1:28:3,1:29:3,1:31:3 timeout 900 python bench.py --no-api --no-cpu-baseline --no-f32 --steps 100
run f64_occ4 GD_OCCUPANCY=1:21:4,1:24:4,1:25:4 timeout 900 python bench.py --no-api --no-cpu-baseline --no-f32 --steps 100
run f64_gch8 GD_HIPCC_EXTRA=-DGD_OC_GCH=8 timeout 900 python bench.py --no-api --no-cpu-baseline --no-f32 --steps 100
BENCH_ARGS="--gradient" bash scripts/pmc_quick.sh 2>&1 | tail -14
BENCH_ARGS="" bash scripts/pmc_quick.sh 2>&1 | tail -14
