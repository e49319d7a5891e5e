# fuzzer with isolated nodes and self loops in the graph families, every mode
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 31 32 33 34 35 36; do timeout 2400 python scripts/fuzz_parity.py 50 --seed=$s 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-900; done
