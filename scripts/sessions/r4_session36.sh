# sequential double value + gradient solves with 8 / 12 gathers in flight instead of 16: does the third wave fit then?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for g in 8 12; do echo "GCH=$g"; GD_HIPCC_EXTRA=-DGD_OC_GCH=$g timeout 900 python scripts/oc_sweep.py --f64 --grad --waves=2,3 2>&1 | tail -11; done
