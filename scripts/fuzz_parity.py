#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the dense fp64 oracle: random
graph families (trees, rings with chords, dense weighted graphs, stars, single
edges, one-node graphs mixed in), random sizes up to 36 nodes, random kernel
composites, q, arithmetic and call mode (symmetric / X x Y, nodal, lmin = 1,
diag, value + gradient) -- every round a fresh small problem, every result
held to the oracle.  Round 5: the pair-list API (`pairlist`), `Normalize(
DotProduct())` and `Convolution` microkernels on nodes and edges of sparse and
dense graphs (`features`), spatial graphs of 65-300 nodes for the 16-wave
on-the-fly variants and the streamed / general solvers (`spatial`), and the
pair-sharded path on two ranks (`sharded`).  What the fixed cases of tests/ do not enumerate: odd
sizes next to each other in one job list, degenerate partners, every solver
family meeting in one launch order.

    python scripts/fuzz_parity.py [rounds] [--seed=N] [--modes=sym,retheta,...]
"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import networkx as nx                                               # noqa: E402
import numpy as np                                                  # noqa: E402
from graphdot_amd.graph import Graph                                # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend  # noqa
from graphdot_amd.microkernel import (                              # noqa: E402
    Additive, Constant, Convolution, DotProduct, KroneckerDelta, Normalize,
    RationalQuadratic, SquareExponential, TensorProduct)
from oracle import mgk as oracle                                    # noqa: E402

rounds = int(next((a for a in sys.argv[1:] if not a.startswith('--')), 40))
seed = int(next((a.split('=')[1] for a in sys.argv[1:]
                 if a.startswith('--seed=')), 0))
only_modes = next((a.split('=')[1].split(',') for a in sys.argv[1:]
                   if a.startswith('--modes=')), None)
rng = np.random.default_rng(seed)


def random_graph(kind, weighted):
    r = int(rng.integers(1 << 30))
    if kind == 'tree':
        n = int(rng.integers(2, 30))
        g = nx.random_labeled_tree(n, seed=r) \
            if hasattr(nx, 'random_labeled_tree') else nx.path_graph(n)
    elif kind == 'ring':
        g = nx.newman_watts_strogatz_graph(int(rng.integers(5, 36)),
                                           int(rng.choice([2, 4])), 0.15, seed=r)
    elif kind == 'bigring':                 # the multi-wave slot variants
        g = nx.newman_watts_strogatz_graph(int(rng.integers(36, 64)),
                                           int(rng.choice([4, 6])), 0.1, seed=r)
    elif kind == 'dense':
        n = int(rng.integers(3, 33))
        g = nx.gnp_random_graph(n, float(rng.uniform(0.6, 1.0)), seed=r)
        for u in range(n - 1):
            g.add_edge(u, u + 1)
    elif kind == 'star':
        g = nx.star_graph(int(rng.integers(2, 20)))
    elif kind == 'iso':                     # isolated nodes next to a ring
        n = int(rng.integers(4, 20))
        g = nx.cycle_graph(n)
        g.add_nodes_from(range(n, n + int(rng.integers(1, 4))))
    elif kind == 'selfloop':                # self loops inside a tree
        n = int(rng.integers(3, 20))
        g = nx.random_labeled_tree(n, seed=r) \
            if hasattr(nx, 'random_labeled_tree') else nx.path_graph(n)
        for v in rng.choice(n, size=min(n, 2), replace=False):
            g.add_edge(int(v), int(v))
    elif kind == 'edge':
        g = nx.path_graph(2)
    else:                                   # one node, one self loop
        g = nx.Graph()
        g.add_edge(0, 0)
    for v in g.nodes:
        g.nodes[v]['category'] = int(rng.integers(1, 4))
        g.nodes[v]['radius'] = float(rng.choice([1.0, 1.5, 2.0]))
    for e in g.edges:
        g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0])) if weighted else 1.0
        g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
        g.edges[e]['order'] = int(rng.integers(1, 3))
    return Graph.from_networkx(g, weight='w' if weighted else None)


def feature_graph(kind, weighted):
    """`random_graph` with a fixed-length vector `fp` and a variable-length
    list `bag` on every node and edge: the attributes the `Normalize(
    DotProduct())` and `Convolution` microkernels work on (variable-length
    payloads behind the graph images, no label classes)."""
    r = int(rng.integers(1 << 30))
    if kind == 'dense':
        n = int(rng.integers(3, 24))
        g = nx.gnp_random_graph(n, float(rng.uniform(0.6, 1.0)), seed=r)
        for u in range(n - 1):
            g.add_edge(u, u + 1)
    elif kind == 'ring':
        g = nx.newman_watts_strogatz_graph(int(rng.integers(5, 40)),
                                           int(rng.choice([2, 4])), 0.15, seed=r)
    else:
        n = int(rng.integers(2, 24))
        g = nx.random_labeled_tree(n, seed=r) \
            if hasattr(nx, 'random_labeled_tree') else nx.path_graph(n)
    for v in g.nodes:
        g.nodes[v]['category'] = int(rng.integers(1, 4))
        g.nodes[v]['fp'] = np.round(rng.uniform(0.1, 1.0, size=3), 3).astype(
            np.float32)
        g.nodes[v]['bag'] = rng.integers(1, 4, size=int(rng.integers(1, 4))) \
            .astype(np.int32)
    for e in g.edges:
        g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0])) if weighted else 1.0
        g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
        g.edges[e]['fp'] = np.round(rng.uniform(0.1, 1.0, size=3), 3).astype(
            np.float32)
        g.edges[e]['bag'] = rng.integers(1, 4, size=int(rng.integers(1, 4))) \
            .astype(np.int32)
    return Graph.from_networkx(g, weight='w' if weighted else None)


def feature_kernels():
    node = [TensorProduct(category=KroneckerDelta(float(rng.uniform(0.3, 0.8))),
                          fp=Normalize(DotProduct())),
            TensorProduct(category=KroneckerDelta(0.5),
                          bag=Convolution(KroneckerDelta(
                              float(rng.uniform(0.3, 0.9))))),
            TensorProduct(fp=Normalize(DotProduct()),
                          bag=Convolution(KroneckerDelta(0.5))),
            ][int(rng.integers(3))]
    edge = [TensorProduct(fp=Normalize(DotProduct())),
            TensorProduct(bag=Convolution(KroneckerDelta(
                float(rng.uniform(0.3, 0.9))))),
            TensorProduct(length=SquareExponential(float(rng.uniform(0.5, 2.0))),
                          fp=Normalize(DotProduct())),
            TensorProduct(length=SquareExponential(1.0),
                          bag=Convolution(KroneckerDelta(0.6))),
            ][int(rng.integers(4))]
    return node, edge


def _sharded_worker(rank, world, port, path):
    """One rank of the `sharded` mode: the pickled problem through
    `distributed_backend()` (gloo: the ranks share the GPU), results to
    `<path>.rank<r>.npz`."""
    import pickle
    import torch                                  # noqa: F401  (first)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from graphdot_amd.kernel.marginalized._sharded import distributed_backend
    with open(path, 'rb') as f:
        G, kseed, q, real, kw = pickle.load(f)
    # (the microkernel classes are made by a decorator and do not pickle:
    # both sides draw the kernels of this mode from the same sub-seed)
    kn, ke = random_kernels(np.random.default_rng(kseed))
    be = distributed_backend(device=0, real=real)
    be.rebalance_min_jobs = 1
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=be, **kw)
    K = k(G)
    K2, dK = k(G, eval_gradient=True)
    h = max(1, len(G) // 2)
    Kxy = k(G[:h], G[h:] or G[:1])
    np.savez(f'{path}.rank{rank}.npz', K=K, K2=K2, dK=dK, Kxy=Kxy)
    dist.destroy_process_group()


def random_kernels(rng=rng):
    node = [TensorProduct(category=KroneckerDelta(float(rng.uniform(0.2, 0.8)))),
            TensorProduct(category=KroneckerDelta(0.5),
                          radius=SquareExponential(float(rng.uniform(0.5, 2.0)))),
            TensorProduct(category=KroneckerDelta(0.4),
                          radius=RationalQuadratic(float(rng.uniform(0.5, 2.0)),
                                                   float(rng.uniform(0.5, 3.0)))),
            Additive(category=KroneckerDelta(0.3) * 0.5,
                     radius=SquareExponential(1.2) * 0.5),
            Constant(1.0),
            TensorProduct(radius=SquareExponential(float(rng.uniform(0.5, 1.5))),
                          category=KroneckerDelta(0.5)) ** float(rng.uniform(0.5, 2.5)),
            ][int(rng.integers(6))]
    edge = [TensorProduct(length=SquareExponential(float(rng.uniform(0.3, 2.0)))),
            TensorProduct(order=KroneckerDelta(float(rng.uniform(0.3, 0.9)))),
            TensorProduct(order=KroneckerDelta(0.6),
                          length=SquareExponential(1.0)),
            TensorProduct(length=RationalQuadratic(float(rng.uniform(0.5, 2.0)),
                                                   float(rng.uniform(0.5, 3.0)))),
            Constant(1.0),
            (TensorProduct(length=SquareExponential(float(rng.uniform(0.6, 1.8))))
             * 0.6 + 0.4) ** float(rng.uniform(0.5, 2.5)),
            ][int(rng.integers(6))]
    return node, edge


def check(name, got, want, rtol, atol=0.0):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.isfinite(got).all(), name
    err = np.abs(got - want) - (atol + rtol * np.abs(want))
    if err.max() > 0:
        at = np.unravel_index(np.argmax(err), err.shape)
        print('worst entry', at, 'got', got[at], 'want', want[at],
              'relative', got[at] / want[at] - 1, flush=True)
        print('entries beyond the tolerance:', int((err > 0).sum()), 'of',
              err.size, np.argwhere(err > 0)[:12].tolist(), flush=True)
    assert err.max() <= 0, (name, float(np.abs(got / want - 1).max()))


def main():
    # (double builds are held to 2e-9: the oracle must not evaluate the
    # microkernels in the float32 the frames store the attributes in)
    oracle.WIDE_ROWS = True
    t0 = time.time()
    kinds = ['tree', 'ring', 'dense', 'dense', 'star', 'edge', 'loop', 'iso',
             'selfloop']
    stats = {}
    for it in range(rounds):
        weighted = bool(rng.integers(2))
        family = rng.choice(['mixed', 'dense', 'sparse', 'large'])
        pool = {'mixed': kinds, 'dense': ['dense', 'dense', 'star'],
                'sparse': ['tree', 'ring', 'edge'],
                'large': ['bigring', 'bigring', 'ring', 'tree']}[family]
        G = Graph.unify_datatype([random_graph(rng.choice(pool), weighted)
                                  for _ in range(int(rng.integers(3, 9)))])
        kn, ke = random_kernels()
        q = float(rng.choice([0.01, 0.05, 0.2, 0.5]))
        real = [np.float32, np.float64][int(rng.integers(2))]
        f64 = real is np.float64
        be = HIPBackend(real=real)
        k = MarginalizedGraphKernel(kn, ke, q=q, backend=be,
                                    **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
        # (double: the device keeps the degrees as float32 sums like the
        # reference -- exact for the dyadic weights used here)
        # (float: at q = 0.01 the systems of unlabeled 30-60-node graphs are
        # conditioned like 1 / q -- values of 2e4 came out 2.8e-5 off at the
        # reference's stopping rule)
        rtol = 2e-9 if f64 else 2e-5 * max(1.0, 0.05 / q)
        mode = rng.choice(only_modes or ['sym', 'xy', 'nodal', 'lmin', 'diag',
                                         'diagnodal', 'grad', 'retheta', 'reuse',
                                         'bulk', 'bulkgrad', 'huge', 'maximin',
                                         'gradmodes', 'ringlist', 'nodalgrad',
                                         'startprob', 'pairlist', 'features',
                                         'spatial', 'sharded', 'mfma'])
        stats[(family, mode, 'f64' if f64 else 'f32')] = \
            stats.get((family, mode, 'f64' if f64 else 'f32'), 0) + 1
        tag = f'round {it} seed {seed}: {family} {mode} {real.__name__} q={q} ' \
              f'weighted={weighted} sizes={[len(g.nodes) for g in G]} ' \
              f'{kn!r} {ke!r}'
        try:
            if mode == 'sym':
                check(tag, k(G), oracle.gram(G, kn, ke, q=q), rtol)
            elif mode == 'xy':
                h = max(1, len(G) // 2)
                check(tag, k(G[:h], G[h:]), oracle.gram(G[:h], kn, ke, Y=G[h:], q=q), rtol)
            elif mode == 'nodal':
                ref = oracle.gram(G[:4], kn, ke, q=q, nodal=True)
                check(tag, k(G[:4], nodal=True), ref, rtol, atol=rtol * np.abs(ref).max())
            elif mode == 'lmin':
                check(tag, k(G, lmin=1), oracle.gram(G, kn, ke, q=q, lmin=1), 10 * rtol)
            elif mode == 'diag':
                check(tag, k.diag(G), np.diag(oracle.gram(G, kn, ke, q=q)), rtol)
            elif mode == 'retheta':
                # the training loop: the same graphs again with other
                # hyperparameters on the same backend (cached layout, new kernel
                # arguments), through clone_with_theta as an optimiser does it
                check(tag, k(G), oracle.gram(G, kn, ke, q=q), rtol)
                for _ in range(3):
                    theta = np.clip(k.theta + rng.normal(scale=0.3, size=len(k.theta)),
                                    k.bounds[:, 0] + 1e-3,
                                    np.minimum(k.bounds[:, 1] - 1e-3, -1e-3))
                    k2 = k.clone_with_theta(theta)
                    want = oracle.gram(G, k2.node_kernel, k2.edge_kernel,
                                       p=k2.p.p if hasattr(k2.p, 'p') else 1.0, q=k2.q)
                    check(tag + f' theta={theta.tolist()}', k2(G), want, rtol)
                    if rng.integers(2):
                        K, dK = k2(G, eval_gradient=True)
                        check(tag + ' (gradient call)', K, want, max(rtol, 1e-7))
            elif mode == 'reuse':
                # the same Graph objects through a float and a double backend in
                # turn, whole list and random subsets / blocks: packings are
                # cached per graph and arithmetic, layouts per list
                full = oracle.gram(G, kn, ke, q=q)
                both = {np.float32: k if not f64 else MarginalizedGraphKernel(
                            kn, ke, q=q, backend=HIPBackend(real=np.float32)),
                        np.float64: k if f64 else MarginalizedGraphKernel(
                            kn, ke, q=q, ftol=1e-13, gtol=1e-12,
                            backend=HIPBackend(real=np.float64))}
                for _ in range(5):
                    r_ = [np.float32, np.float64][int(rng.integers(2))]
                    tol = 2e-9 if r_ is np.float64 else 2e-5
                    idx = rng.permutation(len(G))[:int(rng.integers(1, len(G) + 1))]
                    sub = [G[i] for i in idx]
                    if rng.integers(2) or len(idx) < 2:
                        check(tag + f' subset {idx.tolist()} {r_.__name__}',
                              both[r_](sub), full[np.ix_(idx, idx)], tol)
                    else:
                        h = len(idx) // 2
                        check(tag + f' block {idx.tolist()} {r_.__name__}',
                              both[r_](sub[:h], sub[h:]),
                              full[np.ix_(idx[:h], idx[h:])], tol)
            elif mode == 'pairlist':
                # the pair-list API (AltMarginalizedGraphKernel, the reference's
                # experimental/alterantive_mgk/_kernel.py:26-108): one value per
                # requested pair -- repeated pairs, both orders, the diagonal --
                # against the entries of the oracle's matrix; lmin = 1 and the
                # gradient extension
                from graphdot_amd.experimental.alterantive_mgk import \
                    AltMarginalizedGraphKernel
                alt = AltMarginalizedGraphKernel(
                    kn, ke, q=q, backend=be,
                    **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
                m = int(rng.integers(1, 3 * len(G) + 1))
                ij = rng.integers(0, len(G), size=(m, 2))
                full = oracle.gram(G, kn, ke, q=q)
                check(tag + f' pairs {ij.tolist()}', alt(G, ij),
                      full[ij[:, 0], ij[:, 1]], rtol)
                full1 = oracle.gram(G, kn, ke, q=q, lmin=1)
                check(tag + ' (lmin)', alt(G, ij, lmin=1),
                      full1[ij[:, 0], ij[:, 1]], 10 * rtol)
                v, g_ = alt(G, ij, eval_gradient=True)
                Ko, dKo = oracle.gram(G, kn, ke, q=q, eval_gradient=True)
                check(tag + ' (gradient call)', v, Ko[ij[:, 0], ij[:, 1]],
                      max(rtol, 1e-7))
                want = dKo[ij[:, 0], ij[:, 1]][:, np.asarray(alt.active_theta_mask)]
                scale = np.abs(dKo).max(axis=(0, 1))[
                    np.asarray(alt.active_theta_mask)] + 1e-300
                dev = (np.abs(g_ - want).max(axis=0) / scale).max() if len(ij) else 0.0
                assert np.isfinite(g_).all() and dev < (1e-6 if f64 else 4e-3), \
                    (tag, float(dev))
            elif mode == 'mfma':
                # the dense-tile solver on the matrix cores (mgk_mfma.h: float
                # value solves of DENSE pairs of graphs of at most 32 nodes
                # under an edge kernel that ignores the labels): dense graphs
                # of 2-32 nodes mixed with sparse ones and with graphs just
                # beyond the tile (33-40 nodes: they must take another solver),
                # weighted or not, every output mode of a value solve, against
                # the dense oracle; the launches must include the MFMA kernel
                def dense_graph(n):
                    g = nx.gnp_random_graph(n, float(rng.uniform(0.7, 1.0)),
                                            seed=int(rng.integers(1 << 30)))
                    for u in range(n - 1):
                        g.add_edge(u, u + 1)
                    for v in g.nodes:
                        g.nodes[v]['category'] = int(rng.integers(1, 4))
                        g.nodes[v]['radius'] = float(rng.choice([1.0, 1.5, 2.0]))
                    for e in g.edges:
                        g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0])) \
                            if weighted else 1.0
                        g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
                        g.edges[e]['order'] = int(rng.integers(1, 3))
                    return Graph.from_networkx(g, weight='w' if weighted else None)
                sizes = [int(rng.integers(2, 33)) for _ in range(int(rng.integers(3, 9)))] \
                    + [32, int(rng.integers(33, 41))]
                Gm = [dense_graph(n_) for n_ in sizes] + \
                    [random_graph('tree', weighted), random_graph('ring', weighted)]
                Gm = Graph.unify_datatype([Gm[i_] for i_ in rng.permutation(len(Gm))])
                bm = HIPBackend(real=np.float32)
                km = MarginalizedGraphKernel(kn, Constant(float(rng.choice([1.0, 0.7]))),
                                             q=q, backend=bm)
                tagm = tag + f' mfma: sizes {[len(g.nodes) for g in Gm]}'
                rt = 2e-5 * max(1.0, 0.05 / q)
                want = oracle.gram(Gm, kn, km.edge_kernel, q=q)
                check(tagm, km(Gm), want, rt)
                names = [bm.kernel_name(L['variant'], 1, False, L.get('tab', False))
                         for L in bm.last_plan.launches]
                assert any('mfma' in n_ for n_ in names), (tagm, names)
                h = len(Gm) // 2
                check(tagm + ' (X x Y)', km(Gm[:h], Gm[h:]),
                      oracle.gram(Gm[:h], kn, km.edge_kernel, Y=Gm[h:], q=q), rt)
                check(tagm + ' (diag)', km.diag(Gm), np.diag(want), rt)
                check(tagm + ' (lmin)', km(Gm, lmin=1),
                      oracle.gram(Gm, kn, km.edge_kernel, q=q, lmin=1), 10 * rt)
                refn = oracle.gram(Gm[:3], kn, km.edge_kernel, q=q, nodal=True)
                check(tagm + ' (nodal)', km(Gm[:3], nodal=True), refn, rt,
                      atol=rt * np.abs(refn).max())
            elif mode == 'features':
                # Normalize(DotProduct()) over a vector attribute and Convolution
                # over a variable-length one, on nodes and edges (reference:
                # cpp/basekernel/normalize.h:10-23, microkernel/convolution.py):
                # sparse and DENSE weighted graphs (the dense product evaluates
                # the edge kernel on cells without an edge), value, nodal, gradient
                Gf = Graph.unify_datatype(
                    [feature_graph(rng.choice(['dense', 'dense', 'ring', 'tree']),
                                   weighted) for _ in range(int(rng.integers(3, 8)))])
                knf, kef = feature_kernels()
                tag += f' features: sizes {[len(g.nodes) for g in Gf]} {knf!r} {kef!r}'
                kf = MarginalizedGraphKernel(
                    knf, kef, q=q, backend=be,
                    **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
                check(tag, kf(Gf), oracle.gram(Gf, knf, kef, q=q), rtol)
                refn = oracle.gram(Gf[:3], knf, kef, q=q, nodal=True)
                check(tag + ' (nodal)', kf(Gf[:3], nodal=True), refn, rtol,
                      atol=rtol * np.abs(refn).max())
                K, dK = kf(Gf, eval_gradient=True)
                Ko, dKo = oracle.gram(Gf, knf, kef, q=q, eval_gradient=True)
                check(tag + ' (gradient call)', K, Ko, max(rtol, 1e-7))
                dKo = np.asarray(dKo)[:, :, np.asarray(kf.active_theta_mask)]
                assert np.isfinite(dK).all() and dK.shape == dKo.shape, tag
                if dKo.size:
                    scale = np.abs(dKo).max(axis=(0, 1)) + 1e-300
                    dev = (np.abs(dK - dKo).max(axis=(0, 1)) / scale).max()
                    assert dev < (1e-6 if f64 else 4e-3), (tag, float(dev))
            elif mode == 'spatial':
                # protein-like spatial graphs of 65-300 atoms with 6-25 neighbours
                # among small molecules: the 16-wave on-the-fly variants, the
                # streamed solver (mgk_stream.h) and the general solver behind
                # it, every pair against the C restatement (OpenMP), converged
                import cases
                Gs = cases.protein_like_graphs(
                    int(rng.integers(2, 6)), nmin=65, nmax=300,
                    seed=int(rng.integers(1 << 20)))
                Gs += cases.tang2019_graphs(int(rng.integers(2, 6)),
                                            seed=int(rng.integers(1 << 20)))
                Gs = Graph.unify_datatype(Gs)
                kns = TensorProduct(element=KroneckerDelta(float(rng.uniform(0.2, 0.8))))
                kes = TensorProduct(length=SquareExponential(
                    float(rng.choice([0.05, 0.3, 1.0]))))
                tag += f' spatial: sizes {[len(g.nodes) for g in Gs]} {kns!r} {kes!r}'
                ks = MarginalizedGraphKernel(
                    kns, kes, q=q, backend=be,
                    **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
                K = ks(Gs)
                assert np.array_equal(K, K.T), tag
                i_, j_ = np.triu_indices(len(Gs))
                batch = oracle.TensorProductBatch(Gs, kns, kes)
                ref, _ = batch.run(i_, j_, q=q, tol=1e-13, real='f64', omp=True)
                # (double: the device keeps the degrees as float32 sums of the
                # float32 weights like the reference, _octilegraph.py:109-139,
                # the restatement sums them in double -- exact for the dyadic
                # weights of the other modes, 1e-8 of K on tent weights)
                rtol = max(rtol, 1e-7)
                check(tag, K[i_, j_], ref, rtol)
                h = len(Gs) // 2
                check(tag + ' (block)', ks(Gs[h:], Gs[:h]), K[h:, :h],
                      1e-12 if f64 else 4e-6)
                ref_v, ref_g, _ = batch.run_gradient(i_, j_, q=q, real='f64', omp=True)
                K2, dK = ks(Gs, eval_gradient=True)
                check(tag + ' (gradient call)', K2[i_, j_], ref_v, max(rtol, 1e-7))
                got, want = dK[i_, j_, :], ref_g[:, np.asarray(ks.active_theta_mask)]
                scale = np.abs(want).max(axis=0, keepdims=True)
                rt, at = (1e-5, 1e-8) if f64 else (4e-3 * max(1.0, 0.05 / q), 1e-4)
                worst = float(np.max(np.abs(got - want)
                                     / (rt * np.abs(want) + at * scale + 1e-300)))
                assert worst <= 1.0, (tag, worst)
            elif mode == 'sharded':
                # the pair-sharded path: two ranks (processes of their own sharing
                # the GPU, gloo) through distributed_backend() -- values, value +
                # gradient and an X x Y block bit-equal to this process
                import pickle
                import tempfile
                import torch.multiprocessing as mp
                kw = {'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}
                kseed = int(rng.integers(1 << 30))
                kn, ke = random_kernels(np.random.default_rng(kseed))
                tag += f' sharded: {kn!r} {ke!r}'
                k = MarginalizedGraphKernel(kn, ke, q=q, backend=be, **kw)
                want = (k(G),) + k(G, eval_gradient=True)
                h = max(1, len(G) // 2)
                want_xy = k(G[:h], G[h:] or G[:1])
                with tempfile.TemporaryDirectory() as tmp:
                    path = os.path.join(tmp, 'problem.pkl')
                    with open(path, 'wb') as f:
                        pickle.dump((G, kseed, q, real, kw), f)
                    port = 29600 + (os.getpid() + it) % 300
                    mp.spawn(_sharded_worker, args=(2, port, path), nprocs=2,
                             join=True)
                    for r_ in range(2):
                        got = np.load(f'{path}.rank{r_}.npz')
                        assert np.array_equal(got['K'], want[0]), tag
                        assert np.array_equal(got['K2'], want[1]), tag
                        assert np.array_equal(got['dK'], want[2]), tag
                        assert np.array_equal(got['Kxy'], want_xy), tag
                check(tag, want[0], oracle.gram(G, kn, ke, q=q), rtol)
            elif mode == 'startprob':
                # starting probabilities other than the uniform 1: a constant, and
                # an ad-hoc function of the node attributes (Python callable +
                # device expression), value and gradient
                for pp in (float(rng.uniform(0.2, 3.0)),
                           (lambda nodes: np.asarray(nodes['radius'], dtype=float) + 0.5,
                            'n.radius + 0.5f')):
                    kp = MarginalizedGraphKernel(
                        kn, ke, q=q, p=pp, backend=be,
                        **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
                    what = ' (p = %s)' % (pp if isinstance(pp, float) else pp[1])
                    check(tag + what, kp(G), oracle.gram(G, kn, ke, p=kp.p, q=q), rtol)
                    K, dK = kp(G, eval_gradient=True)
                    Ko, dKo = oracle.gram(G, kn, ke, p=kp.p, q=q, eval_gradient=True)
                    check(tag + what + ' gradient call', K, Ko, max(rtol, 1e-7))
                    dKo = np.asarray(dKo)[:, :, np.asarray(kp.active_theta_mask)] \
                        if np.asarray(dKo).shape[-1] == len(kp.active_theta_mask) \
                        else np.asarray(dKo)
                    if dKo.shape == dK.shape:
                        scale = np.abs(dKo).max(axis=(0, 1)) + 1e-300
                        dev = (np.abs(dK - dKo).max(axis=(0, 1)) / scale).max()
                        assert np.isfinite(dK).all() and dev < (1e-6 if f64 else 4e-3), \
                            (tag + what, float(dev))
            elif mode == 'nodalgrad':
                # the nodal Jacobian (central differences of warm-started
                # re-solves inside the launch, template.cu:226-418) against the
                # oracle's restatement with dense solves, at the reference's own
                # bar (test_kernel.py:289) and 1 % of the column scale
                Gn = [g for g in G if len(g.nodes) <= 24][:3] or G[:1]
                # (double: the re-solves converged -- at the default gtol they stop
                # early by design and meet the reference's 5 % bar only)
                kq = MarginalizedGraphKernel(
                    kn, ke, q=q, backend=be,
                    **({'ftol': 1e-13, 'gtol': 1e-11} if f64 else {}))
                R, dR = kq(Gn, nodal=True, eval_gradient=True)
                Ro, dRo = oracle.gram(Gn, kn, ke, q=q, nodal=True,
                                      eval_gradient=True, eps=kq.eps)
                check(tag + ' (nodal value)', R, Ro, max(rtol, 1e-7),
                      atol=max(rtol, 1e-7) * np.abs(Ro).max())
                ref = dRo[:, :, np.asarray(kq.active_theta_mask)]
                assert dR.shape == ref.shape and np.isfinite(dR).all(), tag
                scale = np.abs(ref).max(axis=(0, 1), keepdims=True)
                if f64:
                    dev = float((np.abs(dR - ref) / (1e-3 * scale + 1e-6)).max())
                else:
                    # (float: differences of two float solves at eps = 0.01 carry
                    # 1e-5 of the value -- 2-3 % of a flat column such as a
                    # rational-quadratic alpha; the reference's own bar)
                    dev = float((np.abs(dR - ref)
                                 / (0.05 * np.abs(ref) + 0.05)).max())
                assert dev <= 1.0, (tag, dev)
            elif mode == 'ringlist':
                # molecules with the variable-length atom attribute of
                # Graph.from_rdkit and a Convolution microkernel over it: payloads
                # behind the graph images (frozen_array), no label classes
                import cases
                from graphdot_amd.microkernel import Convolution
                Gm = cases.config3_graphs(int(rng.integers(4, 12)),
                                          seed=int(rng.integers(1 << 20)),
                                          ring_list=True)
                knm = TensorProduct(
                    atomic_number=KroneckerDelta(float(rng.uniform(0.2, 0.8))),
                    ring_list=Convolution(KroneckerDelta(float(rng.uniform(0.3, 0.9)))))
                kem = [TensorProduct(order=SquareExponential(float(rng.uniform(0.3, 1.5)))),
                       TensorProduct(order=KroneckerDelta(0.5),
                                     aromatic=KroneckerDelta(0.7))][int(rng.integers(2))]
                tag += f' ringlist: sizes {[len(g.nodes) for g in Gm]} {knm!r} {kem!r}'
                km = MarginalizedGraphKernel(
                    knm, kem, q=q, backend=be,
                    **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
                check(tag, km(Gm), oracle.gram(Gm, knm, kem, q=q), rtol)
                refn = oracle.gram(Gm[:3], knm, kem, q=q, nodal=True)
                check(tag + ' (nodal)', km(Gm[:3], nodal=True), refn, rtol,
                      atol=rtol * np.abs(refn).max())
                K, dK = km(Gm, eval_gradient=True)
                Ko, dKo = oracle.gram(Gm, knm, kem, q=q, eval_gradient=True)
                check(tag + ' (gradient call)', K, Ko, max(rtol, 1e-7))
                dKo = dKo[:, :, np.asarray(km.active_theta_mask)]
                scale = np.abs(dKo).max(axis=(0, 1)) + 1e-300
                dev = (np.abs(dK - dKo).max(axis=(0, 1)) / scale).max()
                assert np.isfinite(dK).all() and dev < (1e-6 if f64 else 4e-3), \
                    (tag, float(dev))
            elif mode == 'gradmodes':
                # the analytic gradient through the other call shapes: X x Y
                # blocks, lmin = 1, diag
                mask = np.asarray(k.active_theta_mask)

                def planes(got, want, what):
                    want = np.asarray(want)[..., mask]
                    scale = np.abs(want).reshape(-1, want.shape[-1]).max(axis=0) + 1e-300
                    dev = (np.abs(np.asarray(got) - want).reshape(
                        -1, want.shape[-1]).max(axis=0) / scale).max()
                    assert np.isfinite(got).all() and dev < (1e-6 if f64 else 4e-3), \
                        (tag + ' ' + what, float(dev))
                h = max(1, len(G) // 2)
                K, dK = k(G[:h], G[h:] or G[:1], eval_gradient=True)
                Ko, dKo = oracle.gram(G[:h], kn, ke, Y=G[h:] or G[:1], q=q,
                                      eval_gradient=True)
                check(tag + ' (block)', K, Ko, max(rtol, 1e-7))
                planes(dK, dKo, 'block')
                K, dK = k(G, lmin=1, eval_gradient=True)
                Ko, dKo = oracle.gram(G, kn, ke, q=q, lmin=1, eval_gradient=True)
                check(tag + ' (lmin)', K, Ko, 10 * max(rtol, 1e-7))
                planes(dK, dKo, 'lmin')
                d, dd = k.diag(G, eval_gradient=True)
                do, ddo = oracle.diag(G, kn, ke, q=q, eval_gradient=True)
                check(tag + ' (diag)', d, do, max(rtol, 1e-7))
                planes(dd, ddo, 'diag')
            elif mode == 'maximin':
                # the maximin graph distance fused into the owner-computes
                # launches against the host composition on full nodal matrices
                # from the two-stage solvers (float, the metric's arithmetic)
                from graphdot_amd.metric.maximin import MaxiMin
                from graphdot_amd.kernel.marginalized._backend_hip import (
                    VARIANTS, GENERAL)
                fused = MaxiMin(kn, ke, q=q, backend=HIPBackend())
                host = MaxiMin(kn, ke, q=q,
                               backend=HIPBackend(variants=VARIANTS + [GENERAL]))
                Da, Db = fused(G), host(G)
                assert np.isfinite(Da).all() and np.array_equal(Da, Da.T), tag
                dev = float(np.abs(Da - Db).max())
                assert dev <= 5e-4, (tag, dev)
                if len(G) > 3:
                    dev = float(np.abs(fused(G[:2], G[2:]) - Db[:2, 2:]).max())
                    assert dev <= 5e-4, (tag + ' (block)', dev)
            elif mode == 'huge':
                # a few graphs of 70-300 nodes among small ones: the 16-wave
                # variants at their limits, the two-stage and the general solver
                Gb = []
                for _ in range(int(rng.integers(4, 9))):
                    n_ = int(rng.integers(70, 300))
                    kind_ = rng.choice(['tree', 'ring', 'ladder'])
                    r_ = int(rng.integers(1 << 30))
                    g = (nx.random_labeled_tree(n_, seed=r_) if kind_ == 'tree'
                         and hasattr(nx, 'random_labeled_tree')
                         else nx.newman_watts_strogatz_graph(n_, 4, 0.05, seed=r_)
                         if kind_ == 'ring' else nx.ladder_graph(n_ // 2))
                    for v in g.nodes:
                        g.nodes[v]['category'] = int(rng.integers(1, 4))
                        g.nodes[v]['radius'] = float(rng.choice([1.0, 1.5, 2.0]))
                    for e in g.edges:
                        g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0])) if weighted else 1.0
                        g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
                        g.edges[e]['order'] = int(rng.integers(1, 3))
                    Gb.append(Graph.from_networkx(g, weight='w' if weighted else None))
                Gb += [random_graph(rng.choice(kinds), weighted) for _ in range(6)]
                Gb = Graph.unify_datatype(Gb)
                knb = TensorProduct(category=KroneckerDelta(float(rng.uniform(0.2, 0.8))))
                keb = [TensorProduct(length=SquareExponential(float(rng.uniform(0.3, 2.0)))),
                       Constant(1.0)][int(rng.integers(2))]
                tag += f' huge: sizes {[len(g.nodes) for g in Gb]} {knb!r} {keb!r}'
                kb = MarginalizedGraphKernel(
                    knb, keb, q=q, backend=be,
                    **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
                K = kb(Gb)
                assert np.array_equal(K, K.T), tag
                i_, j_ = np.triu_indices(len(Gb))
                ref, _ = oracle.TensorProductBatch(Gb, knb, keb).run(
                    i_, j_, q=q, tol=1e-13, real='f64', omp=True)
                check(tag, K[i_, j_], ref, rtol)
            elif mode == 'bulkgrad':
                # value + dK/dtheta of 40-110 graphs in one evaluation, every pair
                # and plane against the C restatement of compute_duo + derivative
                Gb = Graph.unify_datatype(
                    [random_graph(rng.choice(kinds + ['bigring']), weighted)
                     for _ in range(int(rng.integers(40, 110)))])
                knb = [TensorProduct(category=KroneckerDelta(float(rng.uniform(0.2, 0.8)))),
                       TensorProduct(category=KroneckerDelta(0.5),
                                     radius=SquareExponential(float(rng.uniform(0.5, 2.0)))),
                       ][int(rng.integers(2))]
                keb = [TensorProduct(length=SquareExponential(float(rng.uniform(0.3, 2.0)))),
                       TensorProduct(order=KroneckerDelta(float(rng.uniform(0.3, 0.9)))),
                       ][int(rng.integers(2))]
                tag += f' bulkgrad: {len(Gb)} graphs {knb!r} {keb!r}'
                kb = MarginalizedGraphKernel(knb, keb, q=q, backend=be)
                K, dK = kb(Gb, eval_gradient=True)
                assert np.isfinite(dK).all() and np.array_equal(K, K.T), tag
                assert np.array_equal(dK, dK.transpose(1, 0, 2)), tag
                i_, j_ = np.triu_indices(len(Gb))
                ref_v, ref_g, _ = oracle.TensorProductBatch(Gb, knb, keb).run_gradient(
                    i_, j_, q=q, real='f64', omp=True)
                check(tag, K[i_, j_], ref_v, max(rtol, 1e-7))
                got, want = dK[i_, j_, :], ref_g[:, np.asarray(kb.active_theta_mask)]
                scale = np.abs(want).max(axis=0, keepdims=True)
                rt, at = (1e-5, 1e-8) if f64 else (4e-3 * max(1.0, 0.05 / q), 1e-4)
                bound = rt * np.abs(want) + at * scale + 1e-300
                worst = float(np.max(np.abs(got - want) / bound))
                assert worst <= 1.0, (tag, worst)
            elif mode == 'bulk':
                # a list of 60-200 graphs of every family in one matrix: many
                # variants, merged launches, the native job layout at size --
                # against the C restatement of the solver (OpenMP), converged
                Gb = Graph.unify_datatype(
                    [random_graph(rng.choice(kinds + ['bigring']), weighted)
                     for _ in range(int(rng.integers(60, 200)))])
                knb = [TensorProduct(category=KroneckerDelta(float(rng.uniform(0.2, 0.8)))),
                       TensorProduct(category=KroneckerDelta(0.5),
                                     radius=SquareExponential(float(rng.uniform(0.5, 2.0)))),
                       Constant(1.0)][int(rng.integers(3))]
                keb = [TensorProduct(length=SquareExponential(float(rng.uniform(0.3, 2.0)))),
                       TensorProduct(order=KroneckerDelta(float(rng.uniform(0.3, 0.9)))),
                       Constant(1.0)][int(rng.integers(3))]
                tag += f' bulk: {len(Gb)} graphs {knb!r} {keb!r}'
                kb = MarginalizedGraphKernel(
                    knb, keb, q=q, backend=be,
                    **({'ftol': 1e-13, 'gtol': 1e-12} if f64 else {}))
                K = kb(Gb)
                assert np.array_equal(K, K.T), tag
                i_, j_ = np.triu_indices(len(Gb))
                ref, _ = oracle.TensorProductBatch(Gb, knb, keb).run(
                    i_, j_, q=q, tol=1e-13, real='f64', omp=True)
                check(tag, K[i_, j_], ref, rtol)
            elif mode == 'diagnodal':
                ref = oracle.diag(G[:4], kn, ke, q=q, nodal=True)
                check(tag, k.diag(G[:4], nodal=True), ref, rtol,
                      atol=rtol * np.abs(ref).max())
            else:
                K, dK = k(G, eval_gradient=True)
                Ko, dKo = oracle.gram(G, kn, ke, q=q, eval_gradient=True)
                check(tag, K, Ko, rtol)
                dKo = dKo[:, :, np.asarray(k.active_theta_mask)]
                scale = np.abs(dKo).max(axis=(0, 1)) + 1e-300
                dev = (np.abs(dK - dKo).max(axis=(0, 1)) / scale).max()
                assert np.isfinite(dK).all() and dev < (1e-6 if f64 else 3e-3), \
                    (tag, float(dev))
        except AssertionError:
            print('FAILED', tag, flush=True)
            plan = getattr(be, 'last_plan', None)
            if plan is not None:
                print('launches:', [(be.kernel_name(L['variant'], plan.C), L['count'])
                                    for L in plan.launches], flush=True)
            raise
    print(f'fuzz ok: {rounds} rounds in {time.time() - t0:.0f} s; '
          f'{len(stats)} (family, mode, arithmetic) combinations')


# (the `sharded` mode spawns its ranks: they re-import this file as
# __mp_main__ and must find the worker, not run the campaign)
if __name__ == '__main__':
    main()
