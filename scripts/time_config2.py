#!/usr/bin/env python3
"""Configuration 2 of BASELINE.json (256 weighted random graphs of 8..48
nodes, 32 896 pairs, both kernel variants): solver time per Gram matrix from
device-resident inputs (replayed plan), fp32 and fp64."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.hip import runtime                                # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend  # noqa

G = cases.config2_graphs(256, seed=0)
n = len(G)
i, j = np.triu_indices(n)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
for name, kernels in (('2a', cases.config2a_kernels),
                      ('2b', cases.config2b_kernels)):
    for real in ((np.float64,) if "--f64" in sys.argv else (np.float32,) if "--f32" in sys.argv else (np.float32, np.float64)):
        kn, ke, q = kernels()
        b = HIPBackend(real=real)
        k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
        plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs,
                         np.arange(n + 1, dtype=np.uint32), n, n, k.n_dims,
                         k.traits(symmetric=True))
        for _ in range(3):
            b.launch(plan)
        runtime.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            b.launch(plan)
            runtime.synchronize()
        dt = (time.perf_counter() - t) / 10
        var = sorted({(L['variant'].W, L['variant'].S) for L in plan.launches
                      if L['variant'].W})
        print(f'config {name} {np.dtype(real).name}: {dt * 1e3:7.3f} ms per Gram '
              f'({len(jobs) / dt / 1e6:6.2f} M pairs/s), variants {var}')
        if '--split' in sys.argv:
            for L in plan.launches:
                runtime.synchronize()
                t = time.perf_counter()
                for _ in range(5):
                    runtime.launch(L['fn'], L['grid'], L['threads'],
                                   L['args'], dynamic_lds=L['dynamic_lds'])
                runtime.synchronize()
                dl = (time.perf_counter() - t) / 5
                v = L['variant']
                print(f'    W{v.W} S{v.S} R{v.R}: pairs {L["count"]:6d}  '
                      f'{dl * 1e3:7.3f} ms  {dl / L["count"] * 1e9:9.1f} ns/pair')
