# fuzzer: last campaign of the round over every mode
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 401 402 403 404; do timeout 600 python scripts/fuzz_parity.py 30 --seed=$s > gpurun_out/s61_$s.log 2>&1; grep -v Warning gpurun_out/s61_$s.log | grep -E "worst|entries|launches|FAILED|fuzz ok|abort|HSA|Error" | cut -c1-1000; grep -n "error:" gpurun_out/s61_$s.log | head -2; done
