#!/usr/bin/env python3
"""What the FIRST numpy-in / numpy-out call of a process pays over the first
call of a later fresh backend (bench.py: api_inclusive.first_call_ms against
fresh_backend_call_ms).  Replays bench.py's order: the timed steps go through
HIPBackend.prepare / LaunchSet first (code objects loaded, host library
touched), then the API calls.  Prints the wall time, the call's own timer
report and a cProfile digest of the process-first call and of a later one.
Usage: profile_process_first_call.py [f32|f64] [--no-profile]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend, LaunchSet
real = np.float32 if 'f32' in sys.argv[1:] else np.float64
profile = '--no-profile' not in sys.argv
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()
n = len(G)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
b = HIPBackend(real=real)
k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs, starts, n, n,
                 k.n_dims, k.traits(symmetric=True))
ls = LaunchSet()
for _ in range(5):
    ls.enqueue(plan)
runtime.synchronize()


def forget():
    for g in G:
        for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
            del g.cookie[key]


def one(label, prof):
    forget()
    kk = MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend(real=real))
    pr = cProfile.Profile() if prof else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    kk(G, timing=not prof)
    if pr:
        pr.disable()
    dt = time.perf_counter() - t0
    print(f'== {label}: {1e3 * dt:.2f} ms' + (' (under cProfile)' if prof else ''))
    if pr:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
        print('\n'.join(l for l in s.getvalue().split('\n')[6:] if l.strip())[:4000])


one('process-first API call', profile)
one('second fresh backend', False)
one('third fresh backend', profile)
one('fourth fresh backend', False)
