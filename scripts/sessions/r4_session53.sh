# fuzzer, gradient through X x Y blocks, lmin = 1 and diag
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 101 102 103; do timeout 3000 python scripts/fuzz_parity.py 30 --seed=$s --modes=gradmodes 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-1400; done
