cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_fuzz_gpu.py -q -x 2>&1 | tail -4
