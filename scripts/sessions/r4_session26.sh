# counters of the dense product (packed two-term evaluation)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
BENCH_ARGS="--config tang2019 --dtype f32" bash scripts/pmc_quick.sh 2>&1 | tail -8
