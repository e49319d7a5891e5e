# stress of the round's kernels: dense product on random subsets, stale scalar cache, soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python scripts/dense_stress.py 150 2>&1 | tail -2
timeout 900 python scripts/stale_cache_stress.py 120 2>&1 | tail -3
timeout 900 python scripts/soak.py 2>&1 | tail -3
