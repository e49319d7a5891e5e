# fuzzer, nodal Jacobian mode
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 121 122; do timeout 3000 python scripts/fuzz_parity.py 25 --seed=$s --modes=nodalgrad > gpurun_out/s57_$s.log 2>&1; grep -v Warning gpurun_out/s57_$s.log | grep -E "worst|entries|launches|FAILED|fuzz ok|abort|HSA" | cut -c1-1400; grep -n "error:" gpurun_out/s57_$s.log | head -3; grep -n "instantiation of\|oc_solver<" gpurun_out/s57_$s.log | head -6 | cut -c1-600; done
