# occupancy sweep of the configuration-2 variants on the current kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python scripts/oc_sweep.py --config2 --waves=2,3,4,5 2>&1 | tail -13
timeout 900 python scripts/oc_sweep.py --config2 --f64 --waves=1,2,3,4 2>&1 | tail -13
