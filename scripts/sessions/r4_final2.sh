# second validation pass of the round: GPU suite, smoke, every bench line, profiles of every profiled workload
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/sessions/r4_bench_all.sh 2>&1 | tail -18
bash scripts/profile_all.sh 2>&1 | tail -20
python scripts/profile_first_call.py f64 > gpurun_out/r4_first_call_f64.log 2>&1; head -5 gpurun_out/r4_first_call_f64.log
