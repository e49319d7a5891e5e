# fuzzer, reuse mode: the same Graph objects through a float and a double backend, subsets and blocks
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 41 42 43; do timeout 2400 python scripts/fuzz_parity.py 40 --seed=$s --modes=reuse 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-900; done
