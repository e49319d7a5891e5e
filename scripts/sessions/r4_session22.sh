# first call after: native table-type check, threaded packer, lighter batch members
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python scripts/profile_process_first_call.py f64 --no-profile 2>&1 | grep -v Warning | grep -E "==|transferring|arena|variants|overall" 
GD_HOST_THREADS=1 timeout 300 python scripts/profile_process_first_call.py f64 --no-profile 2>&1 | grep -v Warning | grep -E "==|transferring" 
timeout 600 python bench.py --steps 50 --cpu-seconds 2 > gpurun_out/s22_f64.json 2> gpurun_out/s22_f64.err; tail -1 gpurun_out/s22_f64.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['api_inclusive'])"
