cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -25 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r4_bench_f64.json 2> gpurun_out/r4_bench_f64.err; echo "bench rc=$?"; head -c 600 gpurun_out/r4_bench_f64.json; echo
timeout 900 python bench.py --gpr --fit > gpurun_out/r4_bench_fit32.json 2> gpurun_out/r4_bench_fit32.err; echo "fit rc=$?"; head -c 1500 gpurun_out/r4_bench_fit32.json; echo
timeout 900 python bench.py --gpr --fit --dtype f64 > gpurun_out/r4_bench_fit64.json 2> gpurun_out/r4_bench_fit64.err; echo "fit64 rc=$?"; head -c 1500 gpurun_out/r4_bench_fit64.json; echo
mkdir -p gpurun_out/jit && cp -n graphdot_amd/_jit_cache/*.hsaco gpurun_out/jit/ 2>/dev/null
