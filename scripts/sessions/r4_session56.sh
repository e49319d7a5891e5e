# a longer campaign of the fuzzer over every mode, fresh seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 201 202 203 204 205 206 207 208; do timeout 3000 python scripts/fuzz_parity.py 40 --seed=$s 2>&1 | grep -v Warning | grep -E "worst|entries|launches|FAILED|fuzz ok|Error|error|abort|HSA" | cut -c1-1400; done
