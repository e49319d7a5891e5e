#!/usr/bin/env python3
"""How well does a cost model linear in (product-graph nonzeros, product-graph
rows, 1) -- the shape of `_sharded.predict_cost` -- explain the measured
per-variant launch times of profiles/r02_bench_{f64,f32}.json?  (Host only.)
Answer on MI355X: not well -- residuals of -57 ... +9 %: the time per pair
steps with the solver variant (registers -> waves per SIMD), not with the
arithmetic.  Hence the snake dealing of `_sharded.partition`, which is balanced
whatever the weights are."""
import json
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend  # noqa

for real, name in ((np.float64, 'f64'), (np.float32, 'f32')):
    b = HIPBackend(real=real)
    G = cases.config3_graphs(1000)
    kn, ke, q = cases.config3_kernels()
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    i, j = np.triu_indices(len(G))
    jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(
        np.dtype([('i', np.uint32), ('j', np.uint32)]))
    traits = k.traits(symmetric=True, eval_gradient=False)
    dgraphs, _, jobs2, C, used, order_all, launches, _ = b._frontend(
        G, kn, ke, k.p, jobs, traits)
    n_node = np.array([g.n_node for g in dgraphs])
    n_nz = np.array([g.n_nz for g in dgraphs])
    line = json.loads(open(os.path.join(
        ROOT, 'profiles', f'r02_bench_{name}.json')).read().strip()
        .splitlines()[-1])
    measured = {kk['kernel']: kk for kk in line['kernels']}
    A, t = [], []
    for L in launches:
        idx = order_all[L['offset']:L['offset'] + L['count']]
        ji = jobs2['i'][idx].astype(int)
        jj = jobs2['j'][idx].astype(int)
        nm = b.kernel_name(L['variant'], 1, False, True)
        if nm not in measured:
            continue
        ns = measured[nm]['isolated_ms'] * 1e6
        A.append([(n_nz[ji] * n_nz[jj]).sum(), (n_node[ji] * n_node[jj]).sum(),
                  L['count']])
        t.append(ns)
        print(f'{name} {nm:32s} {L["count"]:7d} pairs  {ns / L["count"]:6.2f} ns/pair')
    A, t = np.array(A, float), np.array(t)
    x = np.linalg.lstsq(A, t, rcond=None)[0]
    print(name, 'least squares (ns per nonzero, per row, per pair):', np.round(x, 4))
    print(name, 'relative residuals per launch:', np.round((A @ x - t) / t, 2))
