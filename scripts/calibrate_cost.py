#!/usr/bin/env python3
"""Measure the cost table of the pair-sharding planner
(graphdot_amd/kernel/marginalized/cost_table.json, read by
_sharded.cost_table): for every solver variant the QM7-like benchmark set
uses, in both arithmetics, value and value + gradient, the launch is cut into
contiguous slices of its cost-sorted job list, every slice is timed as a
launch of its own, and  t_slice - tail_v = a * pairs + b * sum(nnz1 nnz2 + 4 n1 n2)
is fitted per variant (a, b >= 0); tail_v is the measured duration of a
256-pair launch of the variant (one pair's latency: what the ramp and the
drain of every launch cost together).  Needs the GPU.

    python scripts/calibrate_cost.py [--out path] [--tail-us 12]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.hip import runtime                                # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend  # noqa
from graphdot_amd.kernel.marginalized import _sharded               # noqa: E402

out = [a.split('=')[1] for a in sys.argv if a.startswith('--out=')]
out = out[0] if out else _sharded._COST_TABLE_PATH
tl = [a.split('=')[1] for a in sys.argv if a.startswith('--tail-us=')]
tail_us = float(tl[0]) if tl else 12.0
n = 1000
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
table = {'tail_us': tail_us,
         'provenance': 'scripts/calibrate_cost.py on MI355X, QM7-like set '
                       '(1000 molecules, 500 500 pairs)'}


def timed(b, k, traits, local):
    plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, local,
                     starts, n, n, k.n_dims, traits, packed=True)
    assert len(plan.launches) == 1, [L['variant'] for L in plan.launches]
    for L in plan.pre_launches:
        runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                       dynamic_lds=L['dynamic_lds'])
    L = plan.launches[0]
    for _ in range(2):
        runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                       dynamic_lds=L['dynamic_lds'])
    runtime.synchronize()
    reps = 7
    t0 = time.perf_counter()
    for _ in range(reps):
        runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                       dynamic_lds=L['dynamic_lds'])
    runtime.synchronize()
    return 1e9 * (time.perf_counter() - t0) / reps      # ns per launch


for real, f in ((np.float64, 'f64'), (np.float32, 'f32')):
    for grad in (False, True):
        C = 2 if grad else 1
        b = HIPBackend(real=real, min_launch=0)
        k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
        traits = k.traits(symmetric=True, eval_gradient=grad)
        dgraphs, _, jobs2, C_, used, order_all, launches, _ = b._frontend(
            G, kn, ke, k.p, jobs, traits)
        n_node = np.array([g.n_node for g in dgraphs], np.int64)
        n_nz = np.array([g.n_nz for g in dgraphs], np.int64)
        slopes = []
        for L in launches:
            ids = order_all[L['offset']:L['offset'] + L['count']]
            Q = int(max(1, min(4, L['count'] // 16384)))
            A, t = [], []
            for part in np.array_split(ids, Q):
                local = np.ascontiguousarray(jobs[part])
                ji = local['i'].astype(np.int64)
                jj = local['j'].astype(np.int64)
                arith = _sharded.predict_cost(n_node, n_nz, ji, jj)
                A.append([len(part), float(arith.sum())])
                t.append(timed(b, k, traits, local))
            few = np.ascontiguousarray(jobs[ids[:min(256, len(ids))]])
            lat = timed(b, k, traits, few)
            A, t = np.array(A), np.maximum(np.array(t) - lat, 1.0)
            if Q >= 2:
                x = np.linalg.lstsq(A, t, rcond=None)[0]
            else:
                x = np.array([t[0] / A[0, 0], 0.0])
            if x[0] < 0 or x[1] < 0:      # degenerate fit: arithmetic only
                x = np.array([0.0, t.sum() / A[:, 1].sum()])
            key = f'{f}/C{C}/{_sharded.variant_key(L["variant"])}'
            table[key] = [float(x[0]), float(x[1])]
            # what a launch of this variant costs beyond its pairs: the time
            # of a launch too small to fill the chip (one pair's latency --
            # the ramp and the tail of every launch add up to about that)
            table[key + '/tail_us'] = float(lat) * 1e-3
            slopes.append(t.sum() / A[:, 1].sum())
            resid = (A @ x - t) / t
            print(f'{key:44s} {L["count"]:7d} pairs  '
                  f'{t.sum() / A[:, 0].sum():7.2f} ns/pair  a {x[0]:7.3f} '
                  f'b {x[1]:8.5f}  tail {lat * 1e-3:5.1f} us  '
                  f'resid {np.round(resid, 3)}', flush=True)
        table[f'{f}/C{C}/mean_ns_per_arith'] = float(np.mean(slopes))
with open(out, 'w') as fh:
    json.dump(table, fh, indent=1, sort_keys=True)
print('wrote', out)
