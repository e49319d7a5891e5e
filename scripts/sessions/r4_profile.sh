# rocprofv3 kernel stats + PMC passes + traffic counters for every profiled workload (scripts/profile_all.sh), then the bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/profile_all.sh 2>&1 | tail -20
