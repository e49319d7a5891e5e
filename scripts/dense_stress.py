#!/usr/bin/env python3
"""The dense product of the on-the-fly solvers reads cells of p that no row
publishes (weight 0 records against them) and stages both graphs' records in
LDS per pair: evaluate many random subsets and X x Y blocks of a dense
molecular set -- other pair orders, other neighbours in the job list, other
leftovers in LDS -- and hold every result to the matrix of the whole set."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
n, rounds = 160, int(sys.argv[1]) if len(sys.argv) > 1 else 150
G = cases.tang2019_graphs(n, seed=11)
kn, ke, q = cases.tang2019_kernels()
be = HIPBackend(real=np.float32)
k = MarginalizedGraphKernel(kn, ke, q=q, backend=be)
full = k(G)
assert np.isfinite(full).all() and np.array_equal(full, full.T)
full2, dfull = k(G, eval_gradient=True)
assert np.allclose(full2, full, rtol=2e-5) and np.isfinite(dfull).all()
rng = np.random.default_rng(3)
worst = worst_g = 0.0
for r in range(rounds):
    m = int(rng.integers(2, 60))
    idx = rng.choice(n, size=m, replace=False)
    sub = [G[i] for i in idx]
    kk = k if r % 3 else MarginalizedGraphKernel(
        kn, ke, q=q, backend=HIPBackend(real=np.float32))
    if r % 2:
        K = kk(sub)
        ref = full[np.ix_(idx, idx)]
    else:
        h = max(1, m // 2)
        K = kk(sub[:h], sub[h:] or sub[:1])
        ref = full[np.ix_(idx[:h], idx[h:] if m > h else idx[:1])]
    assert np.isfinite(K).all(), r
    worst = max(worst, float(np.abs(K / ref - 1).max()))
    if r % 10 == 0:
        K2, dK = kk(sub, eval_gradient=True)
        assert np.isfinite(dK).all(), r
        dref = dfull[np.ix_(idx, idx)]
        scale = np.abs(dref).max(axis=(0, 1))
        worst_g = max(worst_g, float((np.abs(dK - dref).max(axis=(0, 1)) / scale).max()))
print(f'dense stress: {rounds} rounds, worst relative deviation {worst:.2e} '
      f'(values), {worst_g:.2e} (gradient planes, of their largest entry)')
assert worst < 5e-5 and worst_g < 5e-4
