cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python scripts/profile_process_first_call.py f64 2>&1 | grep -v Warning | head -45
