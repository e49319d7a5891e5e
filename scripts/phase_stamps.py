#!/usr/bin/env python
"""Diagnostic: where does a pair spend its cycles?  Builds the solver with
-DGD_STAMPS (s_memtime around setup / CG loop / epilogue, accumulated per
launch) and prints the shares per solver variant for the bench workload.
Not part of the product or of the timed benchmark."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend

real = np.float64 if '--f64' in sys.argv else np.float32
backend = HIPBackend(hipcc_extra=['-DGD_STAMPS'], record_iterations=True,
                     concurrent=False, real=real)
graphs = cases.config3_graphs(1000)
knode, kedge, q = cases.config3_kernels()
kernel = MarginalizedGraphKernel(knode, kedge, q=q, backend=backend)
n = len(graphs)
i, j = np.triu_indices(n)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
# iters buffer: per-job counters, then (behind nX * nY entries) the u64
# phase accumulators
backend._buffer('iters', 4 * (n * n + 2) + 128)
plan = backend.prepare(graphs, knode, kedge, kernel.p, kernel.q, kernel.eps,
                       kernel.ftol, kernel.gtol, jobs,
                       np.arange(n + 1, dtype=np.uint32), n, n, kernel.n_dims,
                       kernel.traits(symmetric=True,
                                     eval_gradient='--gradient' in sys.argv))
off = 4 * ((n * n + 1) & ~1)
acc = np.zeros(8, dtype=np.uint64)
for L in plan.launches:
    plan.buffers['iters'].upload(acc * 0, offset=off)
    runtime.synchronize()
    runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                   dynamic_lds=L['dynamic_lds'])
    runtime.synchronize()
    out = np.zeros(8, dtype=np.uint64)
    import ctypes
    runtime.check(runtime.lib().gd_memcpy_d2h(
        out.ctypes.data, plan.buffers['iters'].ptr + off, 64, None))
    runtime.synchronize()
    tot = float(out[:3].sum())
    print(backend.kernel_name(L['variant'], plan.C, False, L.get('tab', False)), 'pairs', int(out[3]),
          'cycles/pair', round(tot / max(int(out[3]), 1)),
          'setup %.1f%% loop %.1f%% epilogue %.1f%%' % tuple(
              100 * out[:3] / tot),
          '| setup = stage %.1f%% + slots %.1f%% + rows %.1f%%' % tuple(
              100 * out[4:7] / tot))
