# fuzzer campaigns: bulk, bulk gradient, huge graphs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 71 72 73 74 75 76; do timeout 3000 python scripts/fuzz_parity.py 20 --seed=$s --modes=bulk,bulkgrad 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-1400; done
for s in 81 82 83; do timeout 3000 python scripts/fuzz_parity.py 12 --seed=$s --modes=huge 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-1400; done
