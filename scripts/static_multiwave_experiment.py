#!/usr/bin/env python3
"""Round 6, review item 3: configuration 2 (256 Newman-Watts-Strogatz graphs,
degree up to 8) on STATIC row-batch layouts of several waves per pair
(mgk_oc.h seg_layout with W > 1) against the dynamic layouts -- the same
pairs, the same launches otherwise.  The layouts are the most common
workgroup trip profiles of the set (scripts/config2_trip_profiles.py).

    python scripts/static_multiwave_experiment.py [--f64] [--layouts=W:L,W:L,...]
(without --layouts: the layouts HIPBackend._refine_static makes to measure)
"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.hip import runtime                                # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import (         # noqa: E402
    HIPBackend, LaunchSet, OCStatic, OC_VARIANTS, VARIANTS,
    _LARGE_PAIR_SOLVERS)

real = np.float64 if '--f64' in sys.argv else np.float32
spec = [a.split('=')[1] for a in sys.argv if a.startswith('--layouts=')]
spec = spec[0] if spec else '8:30x16x16,8:30x20x16,8:36x16,4:30x16x16,4:30x20x16,4:30x16,4:36x16,16:36x16'
layouts = [OCStatic(*map(int, item.split(':')[1].split('x')), D=8,
                    W=int(item.split(':')[0])) for item in spec.split(',')]
G = cases.config2_graphs(256)
kn, ke, q = cases.config2b_kernels()
n = len(G)
i, j = np.triu_indices(n)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)


def run(backend, label):
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=backend)
    t = k.traits(symmetric=True)
    plan = backend.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs,
                           starts, n, n, k.n_dims, t)
    ls = LaunchSet()
    for _ in range(3):
        ls.enqueue(plan)
    runtime.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ls.enqueue(plan)
    runtime.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 20
    # every launch alone
    per = []
    for L in plan.launches:
        sub = type('P', (), {})()
        sub.launches, sub.pre_launches = [L], []
        for _ in range(2):
            ls.enqueue(sub, serial=True)
        runtime.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ls.enqueue(sub, serial=True)
        runtime.synchronize()
        per.append((backend.kernel_name(L['variant'], 1, False, L.get('tab', False)),
                    int(L['count']), 1e3 * (time.perf_counter() - t0) / 10))
    print(f'{label}: {ms:.3f} ms per step, {len(plan.launches)} launches')
    for name, count, t_ in sorted(per, key=lambda r: -r[2]):
        print(f'    {name:48s} {count:6d} pairs {t_:7.3f} ms  {1e6 * t_ / count:8.1f} ns/pair')
    return k(G)


explicit = any(a.startswith('--layouts=') for a in sys.argv)
K0 = run(HIPBackend(real=real, multiwave_static=False), 'dynamic layouts only')
K1 = run(HIPBackend(real=real, multiwave_static=layouts if explicit else True),
         'with static layouts ' + (spec if explicit else '(made to measure)'))
print('largest relative difference of the two matrices:',
      float(np.abs(K1 / K0 - 1).max()))
