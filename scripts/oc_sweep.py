#!/usr/bin/env python3
"""Occupancy sweep of the owner-computes variants on the benchmark set:
isolated launch time of every variant for every amdgpu_waves_per_eu target.
    python scripts/oc_sweep.py [--f64] [--grad] [--config2] [--waves=3,4,5] [--precompile]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend, OC_VARIANTS

real = np.float64 if '--f64' in sys.argv else np.float32
grad = '--grad' in sys.argv
if '--config2' in sys.argv:
    n = 256
    G = cases.config2_graphs(n, seed=0)
    kn, ke, q = cases.config2b_kernels()
else:
    n = 1000
    G = cases.config3_graphs(n)
    kn, ke, q = cases.config3_kernels()
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
table = {}
wl = [a.split('=')[1] for a in sys.argv if a.startswith('--waves=')]
wave_list = tuple(int(x) for x in wl[0].split(',')) if wl else \
    ((1, 2, 3, 4) if '--config2' in sys.argv else (2, 3, 4, 5, 6))
if '--precompile' in sys.argv:      # no device: fill the JIT cache only
    for waves in wave_list:
        b = HIPBackend(real=real, occupancy={(v.W, v.S): waves for v in OC_VARIANTS})
        k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
        print(waves, len(b.precompile(G, kn, ke, k.p, jobs, k.traits(
            symmetric=True, eval_gradient=grad))), flush=True)
    sys.exit(0)
for waves in wave_list:
    b = HIPBackend(real=real, occupancy={(v.W, v.S): waves for v in OC_VARIANTS})
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs, starts,
                     n, n, k.n_dims, k.traits(symmetric=True, eval_gradient=grad))
    for L in plan.pre_launches:        # the table kernel
        runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                       dynamic_lds=L['dynamic_lds'])
    for L in plan.launches:
        for _ in range(2):
            runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                           dynamic_lds=L['dynamic_lds'])
        runtime.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                           dynamic_lds=L['dynamic_lds'])
        runtime.synchronize()
        table.setdefault(b.kernel_name(L['variant'], plan.C, False, L['tab']), {})[waves] = \
            1e3 * (time.perf_counter() - t0) / 5
for name, row in table.items():
    best = min(row, key=row.get)
    print(f'{name:34s} ' + ' '.join(f'{w}:{t:7.3f}' for w, t in row.items())
          + f'   best {best}')
print('sum of best', sum(min(r.values()) for r in table.values()))
