cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for t in 1 2 4 8; do echo "== GD_HOST_THREADS=$t"; GD_HOST_THREADS=$t python scripts/profile_first_call.py f64 2>&1 | head -16; done > gpurun_out/s7_first_call.log 2>&1
cat gpurun_out/s7_first_call.log
GD_HOST_THREADS=4 python scripts/profile_first_call.py f64 2>&1 | sed -n 17,60p
