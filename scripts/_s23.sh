set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "streamed or large_pair" > gpurun_out/s23_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s23_tests.log | cut -c1-200
run() { name=$1; shift; timeout 600 python bench.py --config large --steps 4 --warmup 1 --no-api --no-f32 --cpu-seconds 1 --no-cpu-baseline "$@" > gpurun_out/s23_$name.json 2> gpurun_out/s23_$name.err
  echo -n "[$(date +%T)] $name rc=$? "; python -c "
import json; d=json.loads(open('gpurun_out/s23_$name.json').read().strip().splitlines()[-1])
print(d['config'].get('pairs'), 'pairs', round(d['ms_per_step'],3), 'ms/step', [(k['kernel'][-14:],k['pairs'],k['grid'],round(k['isolated_ms'],3)) for k in d['kernels']])" 2>&1 | tail -1; }
run l8g --graphs 8 --gradient
run l8g_f64 --graphs 8 --gradient --dtype f64
