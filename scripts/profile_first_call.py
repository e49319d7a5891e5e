#!/usr/bin/env python3
"""First numpy-in / numpy-out call on a fresh backend (what bench.py reports as
api_inclusive.first_call_ms): timings of several trials, the call's own timer
report, and a cProfile of one trial.  Usage: profile_first_call.py [f32|f64]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
real = np.float32 if 'f32' in sys.argv[1:] else np.float64
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()


def fresh():
    for g in G:                      # forget earlier packings (as bench.py does)
        for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
            del g.cookie[key]
    return MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend(real=real))


fresh()(G)                           # code objects loaded, context up
for trial in range(4):
    k = fresh()
    t0 = time.perf_counter()
    k(G)
    t1 = time.perf_counter()
    k(G)
    t2 = time.perf_counter()
    print('trial %d: first call %.2f ms, repeat %.2f ms'
          % (trial, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
fresh()(G, timing=True)
k = fresh()
pr = cProfile.Profile()
pr.enable()
k(G)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumtime').print_stats(35)
st.sort_stats('tottime').print_stats(25)
