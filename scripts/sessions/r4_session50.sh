# fuzzer, bulk gradient mode against the C oracle's compute_duo + derivative
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 61 62 63 64; do timeout 3000 python scripts/fuzz_parity.py 10 --seed=$s --modes=bulkgrad 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-1200; done
