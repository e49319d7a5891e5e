# the extended fuzzer: rational-quadratic and additive kernels, constant kernels, 36-63-node rings, nodal diag
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 11 12 13 14 15 16; do timeout 2400 python scripts/fuzz_parity.py 50 --seed=$s 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error" | cut -c1-700; done
