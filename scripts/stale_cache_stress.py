#!/usr/bin/env python3
"""Job lists and graph headers are read through the scalar cache
(mgk_oc.h scalar_load): evaluate many different subsets of a graph set, each
on a fresh backend -- new arena and job buffers, released and re-allocated at
the same device addresses -- and hold every result to the matrix of the whole
set.  A stale scalar-cache line (headers of the previous arena) would show as
a wrong entry."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
n, rounds = 300, int(sys.argv[1]) if len(sys.argv) > 1 else 300
G = cases.config3_graphs(n, seed=5)
kn, ke, q = cases.config3_kernels()
for real, rtol in ((np.float32, 2e-5), (np.float64, 1e-9)):
    # (double: converged solves -- a pair may run on another solver variant in
    # a subset, and at the default 1e-8 N rule two variants agree to that)
    ftol = 1e-8 if real is np.float32 else 1e-13
    full = MarginalizedGraphKernel(kn, ke, q=q, ftol=ftol,
                                   backend=HIPBackend(real=real))(G)
    rng = np.random.default_rng(1)
    worst = 0.0
    for it in range(rounds):
        m = int(rng.integers(2, 60))
        idx = rng.choice(n, size=m, replace=False)
        sub = [G[i] for i in idx]
        for g in sub:                        # forget the packing: new images
            for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
                del g.cookie[key]
        k = MarginalizedGraphKernel(kn, ke, q=q, ftol=ftol,
                                    backend=HIPBackend(real=real))
        if it % 3 == 0:
            K = k(sub)
            ref = full[np.ix_(idx, idx)]
        elif it % 3 == 1:
            h = m // 2
            K = k(sub[:h], sub[h:])
            ref = full[np.ix_(idx[:h], idx[h:])]
        else:
            K = k.diag(sub)
            ref = full[idx, idx]
        err = float(np.abs(K / ref - 1).max())
        worst = max(worst, err)
        assert err < rtol, (real.__name__, it, m, err)
    print(real.__name__, rounds, 'subsets, worst relative difference', worst)
