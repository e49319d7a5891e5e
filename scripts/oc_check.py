#!/usr/bin/env python3
"""Owner-computes against two-stage solver on the benchmark set: agreement and
time per Gram matrix (device-resident launches).
    python scripts/oc_check.py [--f64] [--grad] [--graphs n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import (
    HIPBackend, VARIANTS, OC_VARIANTS, GENERAL)

real = np.float64 if '--f64' in sys.argv else np.float32
grad = '--grad' in sys.argv
n = int(sys.argv[sys.argv.index('--graphs') + 1]) if '--graphs' in sys.argv else 1000
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
res = {}
for name, variants, tables in (
        ('two-stage', VARIANTS + [GENERAL], False),
        ('owner-computes', OC_VARIANTS + VARIANTS + [GENERAL], False),
        ('oc + tables', OC_VARIANTS + VARIANTS + [GENERAL], 'global')):
    b = HIPBackend(real=real, variants=variants, record_iterations=True,
                   tables=tables)
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs, starts,
                     n, n, k.n_dims, k.traits(symmetric=True, eval_gradient=grad))
    for _ in range(3):
        b.launch(plan)
    runtime.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.launch(plan)
    runtime.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out, g = b.collect(plan)
    it = b.iterations(plan)
    res[name] = (out, g, it)
    print(f'{name:15s} {1e3 * dt:8.3f} ms  {len(jobs) / dt / 1e6:7.1f} M pairs/s  '
          f'mean iterations {it.mean():.2f}')
    for L in plan.launches:
        t0 = time.perf_counter()
        for _ in range(5):
            runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                           dynamic_lds=L['dynamic_lds'])
        runtime.synchronize()
        print(f"     {b.kernel_name(L['variant'], plan.C, False, L['tab']):36s} {L['count']:7d} pairs "
              f"{1e3 * (time.perf_counter() - t0) / 5:7.3f} ms  waves/eu {b.waves_per_eu(L['variant'], plan.C)}")
a = res['two-stage']
for other in ('owner-computes', 'oc + tables'):
    b_ = res[other]
    print(other, ': max rel diff of values', np.max(np.abs(a[0] / b_[0] - 1)))
    if grad:
        ga, gb = a[1].reshape(-1, n * n), b_[1].reshape(-1, n * n)
        sc = np.abs(ga).max(axis=1, keepdims=True)
        print('   max gradient diff / plane scale', np.max(np.abs(ga - gb) / sc))
    print('   iterations differ in', int(np.count_nonzero(a[2] != b_[2])), 'pairs')
