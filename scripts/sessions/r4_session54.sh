# fuzzer, variable-length attributes under a Convolution microkernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 111 112; do timeout 3000 python scripts/fuzz_parity.py 25 --seed=$s --modes=ringlist 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-1400; done
