#!/usr/bin/env python3
"""End-to-end wall time of MarginalizedGraphKernel.__call__ on the QM7-like
set (host work included): first call, repeated calls, repeated calls with a
new theta (the GPR training pattern)."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
mlgk = MarginalizedGraphKernel(kn, ke, q=q)
for label, kw in (('value', {}), ('gradient', {'eval_gradient': True})):
    for rep in range(4):
        if rep == 3:
            mlgk.theta = mlgk.theta + 0.01
        t = time.perf_counter()
        out = mlgk(G, **kw)
        dt = time.perf_counter() - t
        print(f'{label:9s} call {rep}{" (new theta)" if rep == 3 else ""}: '
              f'{dt * 1e3:8.1f} ms')
t = time.perf_counter()
d = mlgk.diag(G)
print(f'diag: {(time.perf_counter() - t) * 1e3:.1f} ms')
if '--profile' in sys.argv:
    import cProfile
    import pstats
    cProfile.run('mlgk(G, eval_gradient=True)', '/tmp/api_prof.out')
    pstats.Stats('/tmp/api_prof.out').sort_stats('cumtime').print_stats(35)
