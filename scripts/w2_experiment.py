#!/usr/bin/env python3
"""Experiment: two waves per pair for the heavier molecular pairs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import (
    HIPBackend, VARIANTS, OC_VARIANTS, OCVariant, GENERAL)
real = np.float64 if '--f64' in sys.argv else np.float32
n = 1000
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
base = [v for v in OC_VARIANTS if v.D == 4]
menus = {
    'default': base,
    'W2 above S24': [v for v in base if v.S <= 24] + [OCVariant(2, 12, 2, 4), OCVariant(2, 16, 3, 4), OCVariant(2, 20, 4, 4), OCVariant(2, 24, 5, 4)],
    'W2 above S20': [v for v in base if v.S <= 20] + [OCVariant(2, 12, 2, 4), OCVariant(2, 16, 2, 4), OCVariant(2, 16, 3, 4), OCVariant(2, 20, 4, 4), OCVariant(2, 24, 5, 4)],
    'W2 all': [OCVariant(2, 8, 1, 4), OCVariant(2, 12, 2, 4), OCVariant(2, 16, 2, 4), OCVariant(2, 16, 3, 4), OCVariant(2, 20, 4, 4), OCVariant(2, 24, 5, 4)],
}
ref = None
for name, menu in menus.items():
    b = HIPBackend(real=real, variants=menu + VARIANTS + [GENERAL])
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs, starts,
                     n, n, k.n_dims, k.traits(symmetric=True))
    for _ in range(3):
        b.launch(plan)
    runtime.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.launch(plan)
    runtime.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out, _ = b.collect(plan)
    if ref is None:
        ref = out
    print(f'{name:14s} {1e3 * dt:7.3f} ms {len(jobs) / dt / 1e6:7.1f} M pairs/s  max rel diff {np.max(np.abs(out / ref - 1)):.1e}',
          [(tuple(L["variant"])[:3], L["count"]) for L in plan.launches])
