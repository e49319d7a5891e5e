import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), '..'), os.path.join(os.path.dirname(__file__), '..', 'tests')]
import numpy as np, cases, time
from oracle import mgk
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
G = cases.config3_graphs(10, seed=17)
knode, kedge, q = cases.config3_kernels()
Ro, dRo = mgk.gram(G, knode, kedge, q=q, nodal=True, eval_gradient=True, eps=1e-2)
for real in (np.float32, np.float64):
    fused = HIPBackend(real=real)
    relaunch = HIPBackend(real=real, nodal_gradient_in_kernel=False)
    a = MarginalizedGraphKernel(knode, kedge, q=q, backend=fused)
    b = MarginalizedGraphKernel(knode, kedge, q=q, backend=relaunch)
    Ra, dRa = a(G, nodal=True, eval_gradient=True)
    Rb, dRb = b(G, nodal=True, eval_gradient=True)
    t0 = time.perf_counter(); a(G, nodal=True, eval_gradient=True); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); b(G, nodal=True, eval_gradient=True); tb = time.perf_counter() - t0
    scale = np.abs(dRo).max(axis=(0, 1))
    print(real.__name__, 'in-kernel %.1f ms, relaunches %.1f ms' % (1e3 * ta, 1e3 * tb))
    print('  fused vs relaunch / scale', np.abs(dRa - dRb).max(axis=(0, 1)) / scale)
    print('  fused vs oracle   / scale', np.abs(dRa - dRo).max(axis=(0, 1)) / scale)
    print('  relaunch vs oracle/ scale', np.abs(dRb - dRo).max(axis=(0, 1)) / scale)
