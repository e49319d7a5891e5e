cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python scripts/bisect_double_accuracy.py 2>&1 | grep -v Warning | tail -9
for s in 1 2 3 4; do timeout 1500 python scripts/fuzz_parity.py 60 --seed=$s 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok" | cut -c1-600; done
