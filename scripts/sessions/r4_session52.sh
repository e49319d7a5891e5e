# fuzzer, maximin mode: fused epilogue against the host composition
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 91 92 93; do timeout 3000 python scripts/fuzz_parity.py 30 --seed=$s --modes=maximin 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort" | cut -c1-1400; done
