cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "systems_of_a_few_rows" 2>&1 | tail -2
