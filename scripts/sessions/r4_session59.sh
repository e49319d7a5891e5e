# fuzzer, starting probabilities
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 131 132; do timeout 1500 python scripts/fuzz_parity.py 25 --seed=$s --modes=startprob > gpurun_out/s59_$s.log 2>&1; grep -v Warning gpurun_out/s59_$s.log | grep -E "worst|entries|launches|FAILED|fuzz ok|abort|HSA|Error" | cut -c1-1400; grep -n "error:" gpurun_out/s59_$s.log | head -2; done
