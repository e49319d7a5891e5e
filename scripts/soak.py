#!/usr/bin/env python3
"""Soak: many evaluations with changing hyperparameters, clone_with_theta as
a GPR optimiser does; watches wall time per call and free device memory."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.hip import runtime                                # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa

G = cases.config3_graphs(300)
kn, ke, q = cases.config3_kernels()
k = MarginalizedGraphKernel(kn, ke, q=q)
hip = ctypes.CDLL('libamdhip64.so')


def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20


theta = np.array(k.theta)
rng = np.random.default_rng(0)
times, mem = [], []
for it in range(300):
    kk = k.clone_with_theta(theta + 0.05 * rng.normal(size=len(theta)))
    t = time.perf_counter()
    if it % 3 == 0:
        kk(G, eval_gradient=True)
    elif it % 3 == 1:
        kk(G[:150], G[150:])
    else:
        kk.diag(G)
    times.append(time.perf_counter() - t)
    if it % 50 == 0:
        runtime.synchronize()
        mem.append(free_mb())
        print(f'call {it:4d}: {1e3 * np.mean(times[-30:]):7.2f} ms/call, '
              f'free device memory {mem[-1]:10.1f} MiB')
assert abs(mem[-1] - mem[1]) < 64, 'device memory keeps growing'
print('soak ok')
