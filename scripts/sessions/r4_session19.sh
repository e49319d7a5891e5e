# occupancy sweep of the static layouts on the round-4 kernels (float scalars, sequential solves changed the register needs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python scripts/oc_sweep.py --f64 --waves=2,3,4 2>&1 | tail -12
timeout 600 python scripts/oc_sweep.py --f64 --grad --waves=2,3 2>&1 | tail -12
timeout 600 python scripts/oc_sweep.py --waves=3,4,5,6 2>&1 | tail -12
timeout 600 python scripts/oc_sweep.py --grad --waves=2,3,4 2>&1 | tail -12
