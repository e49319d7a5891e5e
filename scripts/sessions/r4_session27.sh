# first call after the table passes went to numpy's C API
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python scripts/profile_process_first_call.py f64 --no-profile 2>&1 | grep -v Warning | grep -E "==|transferring|arena|variants|overall" 
timeout 300 python scripts/profile_first_call.py f64 2>&1 | head -4
timeout 600 python bench.py --steps 50 --cpu-seconds 2 > gpurun_out/s27_f64.json 2> gpurun_out/s27_f64.err; tail -1 gpurun_out/s27_f64.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['api_inclusive'])"
timeout 600 python bench.py --steps 50 --cpu-seconds 2 --dtype f32 > gpurun_out/s27_f32.json 2> gpurun_out/s27_f32.err; tail -1 gpurun_out/s27_f32.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['api_inclusive'])"
