# fuzzer with powers and affine composites among the kernels; the suite's fuzz test on the new stream
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_fuzz_gpu.py -q -x 2>&1 | tail -3
for s in 501; do timeout 200 python scripts/fuzz_parity.py 20 --seed=$s --modes=sym,grad,xy,retheta > gpurun_out/s62_$s.log 2>&1; grep -v Warning gpurun_out/s62_$s.log | grep -E "worst|entries|launches|FAILED|fuzz ok|abort|HSA|Error" | cut -c1-1000; grep -n "error:" gpurun_out/s62_$s.log | head -2; done
