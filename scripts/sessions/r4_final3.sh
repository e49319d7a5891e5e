# last validation of the round: GPU suite and smoke on the final tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r4c_default.json 2> gpurun_out/r4c_default.err; tail -1 gpurun_out/r4c_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['api_inclusive'], d['roofline']['frac'], d['cpu_baseline']['value'])"
