#!/bin/bash
# Gathers in flight (GD_OC_GCH) of the owner-computes solvers: time per Gram
# matrix on the benchmark set for each setting, both arithmetics.
cd "$GRAFT_REPO_ROOT"
for g in 4 6 8 12 16; do
  for d in "" "--f64"; do
    echo "GCH=$g $d: $(GD_HIPCC_EXTRA="-DGD_OC_GCH=$g" python scripts/oc_check.py $d 2>&1 | grep 'oc + tables')"
  done
done
