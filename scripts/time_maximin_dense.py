#!/usr/bin/env python3
"""MaxiMin distance + gradient of dense molecular graphs (Tang2019 preset):
fused into the on-the-fly solver's launch against the host composition
(2 (n_theta + 1) + 1 nodal value launches + numpy)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.metric.maximin import MaxiMin
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend, VARIANTS, GENERAL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
G = cases.tang2019_graphs(n, seed=3)
knode, kedge, q = cases.tang2019_kernels()
for name, be in (('fused', HIPBackend()), ('host composition', HIPBackend(variants=VARIANTS + [GENERAL]))):
    mm = MaxiMin(knode, kedge, q=q, backend=be)
    for grad in (False, True):
        mm(G, eval_gradient=grad)
        t0 = time.perf_counter()
        for _ in range(3):
            mm(G, eval_gradient=grad)
        print(f'{name:18s} {n} graphs, gradient {grad}: {1e3 * (time.perf_counter() - t0) / 3:8.1f} ms')
