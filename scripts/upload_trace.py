#!/usr/bin/env python3
"""Every DeviceBuffer.upload / download of the process-first API call and of
a later fresh backend's: bytes, milliseconds (same replay as
profile_process_first_call.py)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend, LaunchSet
real = np.float64
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()
n = len(G)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
log = []
for name in ('upload', 'download'):
    orig = getattr(runtime.DeviceBuffer, name)
    def wrap(self, array, *a, _o=orig, _n=name, **k):
        t0 = time.perf_counter()
        r = _o(self, array, *a, **k)
        log.append((_n, np.asarray(array).nbytes, 1e3 * (time.perf_counter() - t0)))
        return r
    setattr(runtime.DeviceBuffer, name, wrap)
oi = runtime.DeviceBuffer.__init__
def init(self, nbytes, _o=oi):
    t0 = time.perf_counter(); _o(self, nbytes)
    log.append(('malloc', int(nbytes), 1e3 * (time.perf_counter() - t0)))
runtime.DeviceBuffer.__init__ = init
if '--skip-prepare' not in sys.argv:
    b = HIPBackend(real=real)
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs, starts, n, n,
                     k.n_dims, k.traits(symmetric=True))
    if '--no-enqueue' not in sys.argv:
        ls = LaunchSet()
        for _ in range(5):
            ls.enqueue(plan)
        runtime.synchronize()
    if '--drop-plan' in sys.argv:
        del plan, b, k
        import gc; gc.collect()
    print('prepare path:', [(a, b_, round(c, 2)) for a, b_, c in log]); log.clear()
for trial in range(3):
    for g in G:
        for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
            del g.cookie[key]
    kk = MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend(real=real))
    t0 = time.perf_counter(); kk(G); dt = 1e3 * (time.perf_counter() - t0)
    print(f'API call {trial}: {dt:.2f} ms', [(a, b_, round(c, 2)) for a, b_, c in log]); log.clear()
