#!/usr/bin/env python3
"""cProfile of the repeated device-resident evaluation a GPR training loop
makes (MarginalizedGraphKernel.device_gram with new hyperparameters)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()
k = MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend())
theta = np.array(k.theta)
for r in range(3):
    k.clone_with_theta(theta + 1e-3 * r).device_gram(G, eval_gradient=True)
runtime.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for r in range(10):
    kk = k.clone_with_theta(theta + 1e-3 * r)
    kk.device_gram(G, eval_gradient=True)
    runtime.synchronize()
pr.disable()
print('per call %.2f ms' % (1e2 * (time.perf_counter() - t0)))
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
