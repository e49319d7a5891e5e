#!/usr/bin/env python3
"""Where the first kernel evaluation on a fresh backend spends its time
(numpy in -> numpy out, 1000 QM7-like graphs): the Timer report of
MarginalizedGraphKernel.__call__ plus a cProfile of the host side.
    python scripts/first_call.py [--f64] [--profile]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend

real = np.float64 if '--f64' in sys.argv else np.float32
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()
runtime.ensure_device()
runtime.DeviceBuffer(1 << 20)          # context, allocator warm
for trial in range(2):
    for g in G:               # forget the packing, keep the row types
        for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
            del g.cookie[key]
    b = HIPBackend(real=real)
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    if '--profile' in sys.argv and trial == 1:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    K = k(G, timing=(trial == 1))
    dt = time.perf_counter() - t0
    if '--profile' in sys.argv and trial == 1:
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)
    print(f'trial {trial}: first call on a fresh backend {1e3 * dt:.1f} ms')
    t0 = time.perf_counter(); k(G); print(f'   repeat call {1e3 * (time.perf_counter() - t0):.1f} ms')
