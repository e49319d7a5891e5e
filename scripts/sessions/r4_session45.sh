# fuzzer, training-loop mode: the same graphs with new hyperparameters on the same backend
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 21 22 23; do timeout 2400 python scripts/fuzz_parity.py 30 --seed=$s --modes=retheta 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error" | cut -c1-900; done
