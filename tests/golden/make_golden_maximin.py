#!/usr/bin/env python
"""Golden vectors that pin MaxiMin to the reference.
Runs ONLY in the build container (needs /root/reference).

The reference's MaxiMin itself runs only on its CUDA backend, so the pin is
made of two reference-derived parts:

* the raw nodal solutions of every graph pair -- unperturbed and at
  exp(log(theta) +- eps) for q and every node / edge hyperparameter, the grid
  of the reference's finite-difference loop (_backend_cuda.py:230-245,
  _backend.cu:252-378) -- computed by the REFERENCE's own CPU solver
  `M3._mlgk` (graphdot/experimental/metric/m3.py:52-106) imported here under
  the shim of make_golden.py (scipy CG tightened to rtol 1e-13);
* the epilogue of the reference's kernel (_backend.cu:100-185 distance and
  hotspot, :190-404 gradient) restated in oracle/maximin.py and applied to
  those solutions, BOTH as the reference computes the gradient (k12 and the
  distance re-read after the finite-difference loop has left the last
  perturbed solve in the buffer, :383) and with the unperturbed solution
  (this repo's default).

maximin.json holds inputs (graphs, kernels, q, eps) and expected outputs
(distance, hotspot, mirrored hotspot, both gradients, nodal self-similarities
and their Jacobian).  tests/test_oracle.py holds oracle/mgk.py + the same
epilogue to it (CPU); tests/test_parity_gpu.py holds the fused HIP epilogue
to it in both modes (GPU).
"""
import copy
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402


def main():
    mg.install_shims()
    sys.path.insert(0, mg.REF)
    sys.path.insert(1, ROOT)
    import networkx as nx
    from graphdot import Graph
    from graphdot.microkernel import (
        KroneckerDelta, SquareExponential, TensorProduct)
    from graphdot.kernel.marginalized.starting_probability import Uniform
    from graphdot.experimental.metric.m3 import M3
    from graphdot.util.iterable import flatten, fold_like
    from oracle import maximin as omm

    rng = np.random.RandomState(11)
    H = []
    for n in (5, 8, 11, 14, 9):
        g = nx.newman_watts_strogatz_graph(
            n, 3, 0.3, seed=int(rng.randint(1 << 30)))
        for i in g.nodes:
            g.nodes[i]['radius'] = float(rng.choice([1.0, 1.5, 2.0]))
            g.nodes[i]['category'] = int(rng.choice([1, 2, 3]))
        for e in g.edges:
            g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0]))
            g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
        H.append(Graph.from_networkx(g, weight='w'))
    H = Graph.unify_datatype(H)
    knode = TensorProduct(radius=SquareExponential(0.5),
                          category=KroneckerDelta(0.5))
    kedge = TensorProduct(length=SquareExponential(1.0))
    q, eps = 0.05, 1e-2
    pstart = Uniform(1.0)

    def solver(kn, ke, qq):
        m = M3.__new__(M3)
        m.q, m.node_kernel, m.edge_kernel = qq, kn, ke
        return m

    def perturbed(kernel, i, delta):
        # reference: CUDABackend.pack_state(diff_grid=True)
        # (_backend_cuda.py:230-245)
        k2 = copy.deepcopy(kernel)
        t = np.log(np.fromiter(flatten(kernel.theta), float))
        t[i] += delta
        k2.theta = fold_like(np.exp(t), kernel.theta)
        return k2

    # the systems of the finite-difference loop, in column order
    # [q, node theta..., edge theta...] (_backend.cu:212-215)
    node_theta = np.fromiter(flatten(knode.theta), float)
    edge_theta = np.fromiter(flatten(kedge.theta), float)
    grid, denom = [], []
    grid.append((solver(knode, kedge, float(np.exp(np.log(q) + eps))),
                 solver(knode, kedge, float(np.exp(np.log(q) - eps)))))
    denom.append(2 * eps * q)
    for i, t in enumerate(node_theta):
        grid.append((solver(perturbed(knode, i, eps), kedge, q),
                     solver(perturbed(knode, i, -eps), kedge, q)))
        denom.append(2 * eps * t)
    for i, t in enumerate(edge_theta):
        grid.append((solver(knode, perturbed(kedge, i, eps), q),
                     solver(knode, perturbed(kedge, i, -eps), q)))
        denom.append(2 * eps * t)
    base = solver(knode, kedge, q)

    def p_of(g):
        p, dp = pstart(g.nodes)
        return np.asarray(p, float), np.atleast_2d(np.asarray(dp, float))

    # nodal self-similarities and their Jacobian per graph
    selfs = []
    for g in H:
        p, dp = p_of(g)
        k, dk = omm.nodal_self(
            base._mlgk(g, g), [s._mlgk(g, g) for s, _ in grid],
            [s._mlgk(g, g) for _, s in grid], denom, p, dp)
        selfs.append((k, dk))

    pairs = []
    for a in range(len(H)):
        for b in range(a, len(H)):
            g1, g2 = H[a], H[b]
            p1, dp1 = p_of(g1)
            p2, dp2 = p_of(g2)
            x0 = base._mlgk(g1, g2)
            xp = [s._mlgk(g1, g2) for s, _ in grid]
            xm = [s._mlgk(g1, g2) for _, s in grid]
            out = {}
            for compat in (True, False):
                D, hot, hot_m, grad = omm.pair_gradient(
                    x0, xp, xm, denom, p1, p2, dp1, dp2, selfs[a][0],
                    selfs[a][1], selfs[b][0], selfs[b][1],
                    reference_compat=compat)
                out['grad_reference' if compat else 'grad_unperturbed'] = grad
            d = omm.node_distance(omm.postproc(x0, p1, p2), selfs[a][0],
                                  selfs[b][0])
            # how far the runner-up of the hotspot is: a tie-break that a
            # float32 solver may legitimately decide the other way
            flat = np.sort(np.abs(d - D).ravel())
            pairs.append({'i': a, 'j': b, 'distance': float(D),
                          'hotspot': hot, 'hotspot_mirrored': hot_m,
                          'runner_up_gap': float(flat[1]) if len(flat) > 1
                          else 1.0,
                          'x0': x0, **out})

    fixture = {
        'provenance': 'tests/golden/make_golden_maximin.py: raw nodal '
                      'solutions from the reference M3._mlgk (shim, CG rtol '
                      '1e-13), epilogue oracle/maximin.py (_backend.cu:100-404)',
        'graphs': [mg.graph_to_dict(g) for g in H],
        'knode': repr(knode), 'kedge': repr(kedge), 'q': q, 'eps': eps,
        'p': 1.0,
        'columns': ['p', 'q'] + [f'node[{i}]' for i in range(len(node_theta))]
        + [f'edge[{i}]' for i in range(len(edge_theta))],
        'node_theta': node_theta, 'edge_theta': edge_theta,
        'nodal_self': [{'k': k, 'dk': dk} for k, dk in selfs],
        'pairs': pairs,
    }
    with open(os.path.join(HERE, 'maximin.json'), 'w') as f:
        json.dump(mg.jsonable(fixture), f)
    print('maximin.json:', len(pairs), 'pairs; max |compat - unperturbed| / '
          'max |grad| =', max(
              np.abs(np.array(p_['grad_reference'])
                     - np.array(p_['grad_unperturbed'])).max()
              / (np.abs(np.array(p_['grad_unperturbed'])).max() + 1e-30)
              for p_ in pairs))


if __name__ == '__main__':
    main()
