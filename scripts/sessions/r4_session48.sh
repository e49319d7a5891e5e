# fuzzer, bulk mode: 60-200 graphs of every family in one matrix against the C oracle
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 51 52 53 54 55 56; do timeout 3000 python scripts/fuzz_parity.py 12 --seed=$s --modes=bulk 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-1200; done
