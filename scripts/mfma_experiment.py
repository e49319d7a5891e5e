#!/usr/bin/env python3
"""Round-5 experiment: does a dense-tile (MFMA) formulation of the
product-graph mat-vec pay?  (north star: "MFMA only if a dense-tile
formulation of the product-graph SpMV proves profitable"; the reference's own
dense path: graphdot/cpp/marginalized_kernel.h:283-328.)

Workload: the dense molecular graphs of `bench.py --config tang2019` (256
from_ase-like graphs, <= 23 atoms, adjacency 88 % dense, tent weights) with the
preset's node kernel and a SEPARABLE edge kernel -- a constant: one label
class.  Three solvers on the same 32 896 pairs, float:

  product   HIPBackend as shipped: the on-the-fly dense product of mgk_oc.h
            (the edge microkernel per term and iteration on the vector pipe)
  mfma      scripts/mfma_experiment.hip: Y = W1 P W2^T as two 32x32x32 products
            of v_mfma_f32_32x32x2_f32 per CG iteration, every CG vector in the
            accumulator layout, one wave per pair
  oracle    the C restatement (a sample), for the values

and, for scale, the product on the preset's own NON-separable edge kernel
(SquareExponential on the bond length), which no dense-tile form covers.

    python scripts/mfma_experiment.py [--graphs 256] [--steps 20]
"""
import json
import os
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.hip import jit, runtime                           # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import (         # noqa: E402
    HIPBackend, LaunchSet)
from graphdot_amd.microkernel import (                              # noqa: E402
    Constant, KroneckerDelta, TensorProduct)


def arg(name, default):
    for k, a in enumerate(sys.argv):
        if a == name:
            return type(default)(sys.argv[k + 1])
    return default


n_graphs, steps = arg('--graphs', 256), arg('--steps', 20)
G = cases.tang2019_graphs(n_graphs)
h_node, e_const, q = 0.2, 1.0, 0.01
knode = TensorProduct(element=KroneckerDelta(h_node))
kedge = Constant(e_const)
n = len(G)
ii, jj = np.triu_indices(n)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
jobs = np.column_stack((ii, jj)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)


def time_product(kn, ke):
    be = HIPBackend(real=np.float32, record_iterations=True)
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=be)
    plan = be.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs, starts,
                      n, n, k.n_dims, k.traits(symmetric=True))
    ls = LaunchSet()
    for _ in range(3):
        ls.enqueue(plan)
    runtime.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ls.enqueue(plan)
    runtime.synchronize()
    dt = (time.perf_counter() - t0) / steps
    K, _ = be.collect(plan)
    names = [(be.kernel_name(L['variant'], 1, False, L.get('tab', False)),
              L['count'], bool(L.get('dense'))) for L in plan.launches]
    return dt, np.array(K).reshape(n, n, order='F'), names, \
        float(be.iterations(plan).mean())


# ---- the product, separable and non-separable edge kernel ---------------------
dt_prod, K_prod, launches, it_prod = time_product(knode, kedge)
kn_t, ke_t, _ = cases.tang2019_kernels()
dt_se, _, launches_se, it_se = time_product(kn_t, ke_t)

# ---- the MFMA kernel ---------------------------------------------------------------
rec_t = np.dtype([('W', np.float32, (32, 32)), ('deg', np.float32, 32),
                  ('label', np.int32, 32), ('n', np.int32),
                  ('pad', np.int32, 31)])
recs = np.zeros(n, dtype=rec_t)
for k_, g in enumerate(G):
    m = len(g.nodes)
    assert m <= 32
    order = np.argsort(np.asarray(g.nodes['!i']))
    lab = np.asarray(g.nodes['element'])[order]
    ei, ej = np.asarray(g.edges['!i']), np.asarray(g.edges['!j'])
    w = np.asarray(g.edges['!w'], dtype=np.float32)
    W = np.zeros((32, 32), dtype=np.float32)
    W[ei, ej] = w
    W[ej, ei] = w
    deg = W.sum(axis=1, dtype=np.float32)
    deg[deg == 0] = 1
    deg[m:] = 1
    recs[k_]['W'], recs[k_]['deg'], recs[k_]['n'] = W, deg, m
    recs[k_]['label'][:m] = lab
with open(os.path.join(ROOT, 'scripts', 'mfma_experiment.hip')) as f:
    mod = runtime.Module(jit.load_image(jit.compile_source(f.read())))
fn = mod.function('mgk_mfma_dense')
b_graphs = runtime.DeviceBuffer(recs.nbytes)
b_graphs.upload(recs.view(np.uint8))
b_jobs = runtime.DeviceBuffer(jobs.nbytes)
b_jobs.upload(jobs.view(np.uint32))
b_out = runtime.DeviceBuffer(4 * len(jobs))
b_it = runtime.DeviceBuffer(4 * len(jobs))
args_t = np.dtype([('graphs', np.uint64), ('jobs', np.uint64), ('out', np.uint64),
                   ('iters', np.uint64), ('n_jobs', np.uint32), ('q', np.float32),
                   ('e', np.float32), ('h', np.float32), ('ftol', np.float32)],
                  align=True)
a = np.zeros((), dtype=args_t)
a['graphs'], a['jobs'], a['out'], a['iters'] = b_graphs.ptr, b_jobs.ptr, \
    b_out.ptr, b_it.ptr
a['n_jobs'], a['q'], a['e'], a['h'], a['ftol'] = len(jobs), q, e_const, h_node, 1e-8
runtime.synchronize()
ev = [(runtime.Event(), runtime.Event()) for _ in range(steps)]
for _ in range(3):
    runtime.launch(fn, len(jobs), 64, a.tobytes())
runtime.synchronize()
t0 = time.perf_counter()
for k_ in range(steps):
    ev[k_][0].record()
    runtime.launch(fn, len(jobs), 64, a.tobytes())
    ev[k_][1].record()
runtime.synchronize()
dt_mfma = (time.perf_counter() - t0) / steps
ms_mfma = float(np.mean([e0.elapsed_ms(e1) for e0, e1 in ev]))
out = np.empty(len(jobs), dtype=np.float32)
b_out.download(out.view(np.uint8))
its = np.empty(len(jobs), dtype=np.uint32)
b_it.download(its.view(np.uint8))

# ---- values --------------------------------------------------------------------------
from oracle import mgk as oracle                                    # noqa: E402
rng = np.random.default_rng(0)
probe = rng.choice(len(jobs), size=min(60, len(jobs)), replace=False)
ref = np.array([oracle.gram([G[a_]], knode, kedge, Y=[G[b_]], q=q).item()
                for a_, b_ in zip(ii[probe], jj[probe])])
line = {
    'workload': f'{n} dense from_ase-like graphs ({len(jobs)} pairs), node '
                f'kernel KroneckerDelta({h_node}) on the element, edge kernel '
                f'Constant({e_const}) (one label class), q = {q}, float',
    'product_on_the_fly': {
        'ms_per_step': 1e3 * dt_prod, 'pairs_per_s': len(jobs) / dt_prod,
        'launches': launches, 'mean_cg_iterations': it_prod},
    'mfma_dense_tile': {
        'ms_per_step': 1e3 * dt_mfma, 'kernel_ms': ms_mfma,
        'pairs_per_s': len(jobs) / dt_mfma,
        'mean_cg_iterations': float(its.mean()),
        'max_rel_diff_vs_product': float(np.max(np.abs(
            out / K_prod[ii, jj] - 1))),
        'max_rel_diff_vs_oracle_sample': float(np.max(np.abs(
            out[probe] / ref - 1))),
        'mfma_per_iteration': 32, 'registers': mod.attributes(
            'mgk_mfma_dense')},
    'speedup_mfma_over_product': dt_prod / dt_mfma,
    'product_on_the_preset_edge_kernel': {
        'note': 'SquareExponential(0.05) on the bond length: not separable, '
                'no dense-tile form', 'ms_per_step': 1e3 * dt_se,
        'pairs_per_s': len(jobs) / dt_se, 'launches': launches_se,
        'mean_cg_iterations': it_se},
}
print(json.dumps(line))
