#!/usr/bin/env python3
"""cProfile of a small repeated call through the numpy API: the reference's
harness shape (benchmark/kernel/marginalized/time_kernel.py: 48-node
Newman-Watts-Strogatz graphs, batch 1 / 16): where do the 0.3 ms go?
    python scripts/profile_small_call.py [batch]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = cases.nws48_graphs(batch)
kn, ke, q = cases.nws48_kernels()
k = MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend())
for r in range(5):
    k(G)
t0 = time.perf_counter()
for r in range(200):
    k(G)
print('batch %d: %.3f ms per call' % (batch, 5 * (time.perf_counter() - t0)))
k(G, timing=True)
pr = cProfile.Profile()
pr.enable()
for r in range(200):
    k(G)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
