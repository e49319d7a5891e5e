# fuzzer: every mode (nodal Jacobian, ring lists, maximin, bulk ... included now), fresh seeds, final tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 301 302 303 304 305 306; do timeout 1500 python scripts/fuzz_parity.py 36 --seed=$s > gpurun_out/s58_$s.log 2>&1; grep -v Warning gpurun_out/s58_$s.log | grep -E "worst|entries|launches|FAILED|fuzz ok|abort|HSA" | cut -c1-1400; grep -n "error:" gpurun_out/s58_$s.log | head -2; done
