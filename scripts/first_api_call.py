#!/usr/bin/env python3
"""The first numpy-in / numpy-out call of a process whose code objects are
already loaded (what bench.py's api_inclusive.first_call_ms times): a plan is
prepared and run through the device interface first, like the timed region of
bench.py, then a fresh backend's first `kernel(graphs)` is profiled.
    python scripts/first_api_call.py [--f64]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend

real = np.float64 if '--f64' in sys.argv else np.float32
n = 1000
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
b0 = HIPBackend(real=real)
k0 = MarginalizedGraphKernel(kn, ke, q=q, backend=b0)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
plan = b0.prepare(G, kn, ke, k0.p, k0.q, k0.eps, k0.ftol, k0.gtol, jobs, starts,
                  n, n, k0.n_dims, k0.traits(symmetric=True))
b0.launch(plan)
runtime.synchronize()
for g in G:
    for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
        del g.cookie[key]
b = HIPBackend(real=real)
k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
pr = cProfile.Profile()
t0 = time.perf_counter()
if '--profile' in sys.argv:
    pr.enable()
K = k(G, timing='--profile' not in sys.argv)
if '--profile' in sys.argv:
    pr.disable()
print(f'first API call {1e3 * (time.perf_counter() - t0):.1f} ms')
if '--profile' in sys.argv:
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
t0 = time.perf_counter(); k(G); print(f'repeat {1e3 * (time.perf_counter() - t0):.1f} ms')
