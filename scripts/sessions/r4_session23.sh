cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "dense_product_at_the_row_limit or dense_graphs_take or mixed_degree" 2>&1 | tail -5
