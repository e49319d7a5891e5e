cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python scripts/profile_first_call.py f64 > gpurun_out/s11_first_call.log 2>&1; head -60 gpurun_out/s11_first_call.log
python scripts/profile_first_call.py f32 2>&1 | head -5
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "dense or on_the_fly or maximin or molecular or repeated or layout" 2>&1 | tail -3
