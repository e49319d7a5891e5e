# differential fuzzing against the oracle: random graph families, kernels, call modes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 1 2 3; do timeout 1500 python scripts/fuzz_parity.py 60 --seed=$s 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok" | cut -c1-600; done
