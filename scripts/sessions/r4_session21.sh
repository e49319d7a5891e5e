cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "-- no enqueue"; timeout 300 python scripts/upload_trace.py --no-enqueue 2>&1 | grep -v Warning | grep "API call 0" | cut -c1-120
echo "-- enqueue, plan dropped before the call"; timeout 300 python scripts/upload_trace.py --drop-plan 2>&1 | grep -v Warning | grep "API call 0" | cut -c1-120
echo "-- as bench"; timeout 300 python scripts/upload_trace.py 2>&1 | grep -v Warning | grep "API call 0" | cut -c1-120
