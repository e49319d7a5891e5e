#!/usr/bin/env python
"""Time the reference's own CPU statement of the hot path, `M3._mlgk`
(graphdot/experimental/metric/m3.py:52-106: scipy-sparse Kronecker assembly +
scipy.sparse.linalg.cg with a Jacobi preconditioner, one pair per call), on a
seeded sample of the QM7-like benchmark set -- BASELINE.md section 3, item 1.

Runs ONLY in the build container (needs /root/reference, imported under the
same scratch shim as make_golden.py but with scipy's CG left at the
tolerances the reference passes).  Writes profiles/r01_reference_python_cpu.json;
nothing of the reference is copied.
"""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, os.path.join(ROOT, 'tests'), ROOT]

import make_golden                                       # noqa: E402
import scipy.sparse.linalg as spla                       # noqa: E402

_cg = spla.cg
make_golden.install_shims()
spla.cg = _cg                       # the reference's own tolerances
sys.path.insert(0, make_golden.REF)
from graphdot import Graph as RefGraph                                   # noqa
from graphdot.microkernel import (                                      # noqa
    KroneckerDelta, SquareExponential, TensorProduct)
from graphdot.experimental.metric.m3 import M3                          # noqa
import cases                                                            # noqa

n_graphs, n_pairs = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(7165)
mols = [cases.qm7_like_molecule(rng) for _ in range(n_graphs)]
for m in mols:                       # M3 wants weights: unit weights change nothing
    for e in m.edges:
        m.edges[e]['w'] = 1.0
G = RefGraph.unify_datatype([RefGraph.from_networkx(m, weight='w')
                             for m in mols])
knode = TensorProduct(atomic_number=KroneckerDelta(0.5),
                      hcount=SquareExponential(1.0),
                      aromatic=KroneckerDelta(0.8))
kedge = TensorProduct(order=SquareExponential(0.5),
                      conjugated=KroneckerDelta(0.5))
m3 = M3.__new__(M3)
m3.q, m3.node_kernel, m3.edge_kernel = 0.01, knode, kedge

pick = np.random.default_rng(1)
ii = pick.integers(0, n_graphs, n_pairs)
jj = pick.integers(0, n_graphs, n_pairs)
m3._mlgk(G[0], G[1])                                       # warm-up
t0 = time.perf_counter()
vals = [float(np.sum(m3._mlgk(G[a], G[b]))) for a, b in zip(ii, jj)]
dt = time.perf_counter() - t0

# the same pairs through this repo's C restatement (sanity: same quantity)
from oracle import mgk                                                  # noqa
own = cases.config3_graphs(n_graphs)
kn, ke, q = cases.config3_kernels()
batch = mgk.TensorProductBatch(own, kn, ke)
ref, _ = batch.run(ii, jj, q=q, real='f64', tol=1e-13)
rel = float(np.max(np.abs(np.array(vals) / ref - 1)))

out = {
    'what': 'reference CPU path M3._mlgk (graphdot/experimental/metric/'
            'm3.py:52-106), one pair per call, 1 core',
    'value': n_pairs / dt, 'unit': 'graph-pairs/s', 'pairs': n_pairs,
    'seconds': dt, 'cores': 1,
    'sample': f'{n_pairs} random pairs of the first {n_graphs} QM7-like '
              'molecules (seed 7165), config-3 microkernels, q = 0.01',
    'max_rel_diff_vs_c_restatement': rel,
    'host': platform.processor() or platform.machine(),
    'python': platform.python_version(),
    'where': 'build container (the reference does not travel to the GPU box)',
}
print(json.dumps(out, indent=1))
with open(os.path.join(ROOT, 'profiles',
                       'r01_reference_python_cpu.json'), 'w') as f:
    json.dump(out, f, indent=1)
