#!/usr/bin/env python3
"""Configuration 5 of BASELINE.json on one GPU: wall time of one
hyperparameter-fit step (log marginal likelihood + gradient) of a Gaussian
process on the QM7-like set -- kernel + dK/dtheta on the HIP path, dense
algebra through torch on the same GPU."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                  # noqa: E402
from graphdot_amd.model.gaussian_process import GaussianProcessRegressor  # noqa
import cases                                                        # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend  # noqa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
real = np.float64 if '--f64' in sys.argv else np.float32
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
kernel = MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend(real=real))
rng = np.random.default_rng(0)
y = rng.normal(size=n)
d = kernel.diag(G)
gpr = GaussianProcessRegressor(kernel, alpha=float(1e-2 * d.mean()),
                               normalize_y=True)
gpr.X, gpr.y = G, y
theta = np.array(kernel.theta)
for rep in range(4):
    t = time.perf_counter()
    val, grad = gpr.log_marginal_likelihood(theta + 0.01 * rep,
                                            eval_gradient=True)
    dt = time.perf_counter() - t
    print(f'step {rep}: {dt * 1e3:7.1f} ms   kernel {gpr.last_timing["kernel"] * 1e3:6.1f} ms'
          f'   dense algebra ({gpr._dense().device.type}) '
          f'{gpr.last_timing["linalg"] * 1e3:6.1f} ms   logP {val:.6g}')
