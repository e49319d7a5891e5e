cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python scripts/profile_first_call.py f64 > gpurun_out/s6_first_call.log 2>&1; head -150 gpurun_out/s6_first_call.log
