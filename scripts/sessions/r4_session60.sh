# fuzzer: more seeds for the modes with the fewest rounds so far
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for m in nodalgrad ringlist huge bulkgrad reuse; do timeout 700 python scripts/fuzz_parity.py 16 --seed=4${#m}7 --modes=$m > gpurun_out/s60_$m.log 2>&1; echo -n "$m: "; grep -v Warning gpurun_out/s60_$m.log | grep -E "worst|entries|launches|FAILED|fuzz ok|abort|HSA|Error" | cut -c1-1000; grep -n "error:" gpurun_out/s60_$m.log | head -2; done
