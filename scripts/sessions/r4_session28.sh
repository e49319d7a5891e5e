cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GD_API_TIMING=1 timeout 600 python bench.py --steps 50 --cpu-seconds 2 > gpurun_out/s28_f64.json 2> gpurun_out/s28_f64.err; grep -E " ms on " gpurun_out/s28_f64.err | head -14; tail -1 gpurun_out/s28_f64.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['api_inclusive'])"
