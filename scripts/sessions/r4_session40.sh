# the 5e-8 deviations of the double build at q = 0.01 found by the fuzzer: with double scalars in the iteration?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GD_HIPCC_EXTRA=-DGD_OC_FSCAL=0 timeout 1500 python scripts/fuzz_parity.py 8 --seed=2 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok" | cut -c1-400
GD_HIPCC_EXTRA=-DGD_OC_FSCAL=0 timeout 1500 python scripts/fuzz_parity.py 8 --seed=3 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok" | cut -c1-400
