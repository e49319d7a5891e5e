#!/usr/bin/env python3
"""cProfile of the repeated numpy-in / numpy-out call
(MarginalizedGraphKernel.__call__ on 1000 molecular graphs).
Usage: profile_api_call.py [f32|f64] [grad] [torch]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
if 'torch' in sys.argv[1:]:
    import torch
    torch.zeros(1).cuda()
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
real = np.float32 if 'f32' in sys.argv[1:] else np.float64
grad = 'grad' in sys.argv[1:]
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()
k = MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend(real=real))
for r in range(3):
    k(G, eval_gradient=grad)
pr = cProfile.Profile()
t0 = time.perf_counter()
for r in range(10):
    k(G, eval_gradient=grad)
print('per call %.2f ms (no profiler)' % (1e2 * (time.perf_counter() - t0)))
timing = k(G, eval_gradient=grad, timing=True)
pr.enable()
for r in range(10):
    k(G, eval_gradient=grad)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
