"""The CPU oracle (oracle/) pinned against the golden vectors generated from
the reference's own Python oracles (tests/golden/make_golden.py)."""
import numpy as np
import pytest
from _fixtures import load, graphs_from, kernel_from_repr
from graphdot_amd.microkernel import Constant
from oracle import mgk

MLGK = load('mlgk_cases.json')
FAMILIES = ['unlabeled', 'labeled', 'weighted', 'vario-features']
# dense/pcg64: fp64 restatements; pcg32: the reference's device arithmetic,
# the reference's own tests accept 1e-5 for it (test_kernel.py:214)
TOL = {'dense': 1e-11, 'pcg64': 1e-11, 'pcg32': 1e-5}


@pytest.mark.parametrize('mode', ['dense', 'pcg64', 'pcg32'])
@pytest.mark.parametrize('name', FAMILIES)
def test_self_pairs_vs_reference_MLGK(name, mode):
    case = MLGK[name]
    G = graphs_from(case['graphs'])
    knode = kernel_from_repr(case['knode'])
    kedge = kernel_from_repr(case['kedge'])
    for qi, q in enumerate(case['q']):
        for gi, g in enumerate(G):
            R = mgk.gram([g], knode, kedge, q=q, nodal=True, mode=mode)
            ref = np.array(case['R_nodal'][qi][gi])
            assert np.allclose(R, ref, rtol=TOL[mode], atol=0)
            r = mgk.gram([g], knode, kedge, q=q, mode=mode)[0, 0]
            assert r == pytest.approx(case['R'][qi][gi], rel=TOL[mode])


def test_known_answers_from_survey():
    """values quoted from the reference oracle."""
    R = {name: MLGK[name]['R'] for name in FAMILIES}
    assert R['unlabeled'][0][0] == pytest.approx(452.2613065, rel=1e-9)
    assert R['unlabeled'][3] == pytest.approx([12.0, 12.0], rel=1e-12)
    assert R['labeled'][1] == pytest.approx([9.376686034, 20.84356588],
                                            rel=1e-9)
    assert R['weighted'][1] == pytest.approx([11.71550219, 41.02564103],
                                             rel=1e-9)
    assert R['vario-features'][1] == pytest.approx(
        [6.558041573, 4.741470873], rel=1e-9)


@pytest.mark.parametrize('mode', ['dense', 'pcg32'])
def test_self_loops_vs_reference_MLGK(mode):
    for c in MLGK['self-loops']:
        g = graphs_from([c['graph']])[0]
        r = mgk.gram([g], Constant(1.0), Constant(1.0), q=c['q'],
                     mode=mode)[0, 0]
        assert r == pytest.approx(c['R'], rel=TOL[mode])


@pytest.mark.parametrize('mode', ['dense', 'pcg32'])
def test_cross_pairs_vs_reference_M3(mode):
    """M3._mlgk builds part of its system in float32 (scipy CSC of float32
    weights), so it pins cross pairs to about 1e-6 only."""
    for case in load('m3_cross.json').values():
        G = graphs_from(case['graphs'])
        knode = kernel_from_repr(case['knode'])
        kedge = kernel_from_repr(case['kedge'])
        for a in range(len(G)):
            for b in range(len(G)):
                R = mgk.gram([G[a]], knode, kedge, Y=[G[b]], q=case['q'],
                             nodal=True, mode=mode)
                ref = np.array(case['R_nodal'][a][b])
                assert np.allclose(R, ref, rtol=5e-6, atol=0)


def test_closed_form_unlabeled():
    """K = n1 n2 / (1 - (1-q)^2) for unlabeled graphs without isolated nodes
    (SURVEY.md 8c(2))."""
    import cases
    G = cases.config1_graphs()
    n = np.array([len(g.nodes) for g in G], dtype=float)
    for q in (0.05, 0.5):
        R = mgk.gram(G[:5], Constant(1.0), Constant(1.0), q=q)
        assert np.allclose(R, np.outer(n[:5], n[:5]) / (1 - (1 - q)**2),
                           rtol=1e-12)


@pytest.mark.parametrize('name', ['labeled', 'weighted', 'vario-features'])
def test_analytic_gradient_restatement_vs_finite_differences(name):
    """mgk_derivative (marginalized_kernel.h:806-997 restated) against
    central differences of the dense fp64 oracle in the raw hyperparameters."""
    from graphdot_amd.util.iterable import flatten
    case = MLGK[name]
    G = graphs_from(case['graphs'])
    knode = kernel_from_repr(case['knode'])
    kedge = kernel_from_repr(case['kedge'])
    p, q, h = 1.3, 0.05, 1e-6
    K, dK = mgk.gram(G, knode, kedge, p=p, q=q, eval_gradient=True)

    def val(p_, q_, kn, ke):
        return mgk.gram(G, kn, ke, p=p_, q=q_)

    cols = [(val(p + h, q, knode, kedge) - val(p - h, q, knode, kedge)),
            (val(p, q + h, knode, kedge) - val(p, q - h, knode, kedge))]
    for which, k in (('n', knode), ('e', kedge)):
        t = np.array(list(flatten(k.theta)), float)
        for j in range(len(t)):
            tp, tm = t.copy(), t.copy()
            tp[j] += h
            tm[j] -= h
            kp, km = mgk._with_theta(k, tp), mgk._with_theta(k, tm)
            if which == 'n':
                cols.append(val(p, q, kp, kedge) - val(p, q, km, kedge))
            else:
                cols.append(val(p, q, knode, kp) - val(p, q, knode, km))
    fd = np.stack(cols, -1) / (2 * h)
    assert dK.shape == fd.shape
    assert np.allclose(dK, fd, rtol=1e-6, atol=1e-6 * np.abs(fd).max())


def test_duo_solver_matches_two_single_solves():
    case = MLGK['weighted']
    G = graphs_from(case['graphs'])
    knode = kernel_from_repr(case['knode'])
    kedge = kernel_from_repr(case['kedge'])
    s1, s2 = mgk._side(G[0]), mgk._side(G[1])
    V = mgk.node_table(knode, s1, s2)
    E = mgk.edge_table(kedge, s1, s2)
    px = np.full((s1.n, s2.n), 1.7)
    x, y, _ = mgk.solve_pair(s1, s2, V, E, 0.1, 'pcg64', rhs_extra=px)
    xd, yd, _ = mgk.solve_pair(s1, s2, V, E, 0.1, 'dense', rhs_extra=px)
    assert np.allclose(x, xd, rtol=1e-9) and np.allclose(y, yd, rtol=1e-9)


def test_batched_tensorproduct_path_matches_generic():
    import cases
    G = cases.config3_graphs(12, seed=5)
    knode, kedge, q = cases.config3_kernels()
    batch = mgk.TensorProductBatch(G, knode, kedge)
    i, j = np.triu_indices(len(G))
    got, iters = batch.run(i, j, q=q, real='f64', tol=1e-13)
    ref = mgk.gram(G, knode, kedge, q=q)
    assert np.allclose(got, ref[i, j], rtol=1e-9)
    got32, it32 = batch.run(i, j, q=q, real='f32')
    assert np.allclose(got32, ref[i, j], rtol=1e-5)
    assert it32.min() >= 1 and it32.max() < 100


def test_c_gradient_restatement_vs_dense_oracle():
    """`mgk_gram_tp_grad_f64` (compute_duo + derivative in C, the full-size
    gradient checker and CPU baseline of config 5) against the dense fp64
    statement `pair_value(eval_gradient=True)`: both solve the same two
    systems, the C one with the device's stopping rule (1e-10 * 2N)."""
    import cases
    for graphs, kernels in ((cases.config3_graphs(24, seed=3),
                             cases.config3_kernels()),
                            (cases.config2_graphs(6, nmin=6, nmax=12, seed=1),
                             cases.config2b_kernels()),
                            (cases.config2_graphs(6, nmin=6, nmax=12, seed=2),
                             cases.config2a_kernels())):
        knode, kedge, q = kernels
        batch = mgk.TensorProductBatch(graphs, knode, kedge)
        n = len(graphs)
        ii, jj = np.triu_indices(n)
        ii, jj = ii[::5], jj[::5]
        v, g, it = batch.run_gradient(ii, jj, q=q, real='f64')
        for t, (a, b) in enumerate(zip(ii, jj)):
            R, J = mgk.pair_value(graphs[a], graphs[b], knode, kedge, q=q,
                                  eval_gradient=True, tol=1e-13)
            assert v[t] == pytest.approx(R, rel=1e-7)
            assert np.allclose(g[t], J, rtol=1e-6, atol=1e-7 * np.abs(J).max())
        v32, g32, _ = batch.run_gradient(ii, jj, q=q, real='f32')
        assert np.allclose(v32, v, rtol=2e-5)


def _maximin_from_oracle(fx, reference_compat):
    """Distance, hotspots and gradient of every pair of the maximin fixture
    from this repo's dense oracle (mgk.py) and the epilogue restatement
    (oracle/maximin.py)."""
    from oracle import maximin as omm
    G = graphs_from(fx['graphs'])
    knode, kedge = kernel_from_repr(fx['knode']), kernel_from_repr(fx['kedge'])
    q, eps = fx['q'], fx['eps']

    def raw(g1, g2, kn=knode, ke=kedge, qq=q):
        # p = 1, lmin = 0: the nodal matrix IS the raw solution
        return mgk.pair_value(g1, g2, kn, ke, q=qq, nodal=True, tol=1e-13)[0]

    grid, denom = [(dict(qq=float(np.exp(np.log(q) + eps))),
                    dict(qq=float(np.exp(np.log(q) - eps))))], [2 * eps * q]
    for which, kern in (('kn', knode), ('ke', kedge)):
        theta = mgk._flat_theta(kern)
        for j, t in enumerate(theta):
            tp, tm = theta.copy(), theta.copy()
            tp[j], tm[j] = np.exp(np.log(t) + eps), np.exp(np.log(t) - eps)
            grid.append(({which: mgk._with_theta(kern, tp)},
                         {which: mgk._with_theta(kern, tm)}))
            denom.append(2 * eps * t)
    one = [np.ones(len(g.nodes)) for g in G]
    selfs = [omm.nodal_self(raw(g, g), [raw(g, g, **a) for a, _ in grid],
                            [raw(g, g, **b) for _, b in grid], denom,
                            one[k], one[k][None, :])
             for k, g in enumerate(G)]
    out = []
    for pr in fx['pairs']:
        a, b = pr['i'], pr['j']
        out.append(omm.pair_gradient(
            raw(G[a], G[b]), [raw(G[a], G[b], **u) for u, _ in grid],
            [raw(G[a], G[b], **v) for _, v in grid], denom, one[a], one[b],
            one[a][None, :], one[b][None, :], selfs[a][0], selfs[a][1],
            selfs[b][0], selfs[b][1], reference_compat=reference_compat))
    return selfs, out


def test_maximin_restatement_vs_reference_solutions():
    """tests/golden/maximin.json (raw nodal solutions of the REFERENCE's CPU
    solver M3._mlgk through the restated epilogue of its maximin kernel,
    metric/maximin/_backend.cu:100-404) against the same epilogue on this
    repo's dense oracle: distance, hotspot and mirrored hotspot, gradient in
    both forms -- k12 re-read after the finite-difference loop as the
    reference does (:383), and from the unperturbed solve -- and the nodal
    self-similarities with their Jacobian."""
    fx = load('maximin.json')
    for compat, key in ((True, 'grad_reference'), (False, 'grad_unperturbed')):
        selfs, got = _maximin_from_oracle(fx, compat)
        for (k, dk), ref in zip(selfs, fx['nodal_self']):
            assert np.allclose(k, ref['k'], rtol=2e-5)        # (M3: float32 parts)
            assert np.allclose(dk, ref['dk'], rtol=5e-3,
                               atol=2e-4 * np.abs(ref['dk']).max())
        for (D, hot, hot_m, grad), ref in zip(got, fx['pairs']):
            assert D == pytest.approx(ref['distance'], abs=2e-4)
            if ref['runner_up_gap'] > 1e-4 and ref['i'] != ref['j']:
                assert (hot, hot_m) == (ref['hotspot'],
                                        ref['hotspot_mirrored'])
                want = np.array(ref[key])
                assert np.allclose(grad, want, rtol=2e-2,
                                   atol=2e-3 * np.abs(want).max() + 1e-5)
    # the two forms really differ (by O(eps)), most in the q column
    d = [np.abs(np.array(p['grad_reference']) - np.array(p['grad_unperturbed']))
         for p in fx['pairs'] if p['i'] != p['j']]
    assert 1e-4 < max(x.max() for x in d) < 1e-2
    assert all(x[0] == 0 for x in d)        # starting probability: same form


def test_microkernels_in_float64_on_request():
    """The frames store float attributes as float32, and the reference's
    Python solver -- the source of the golden vectors -- evaluates the
    microkernels in that type (numpy keeps float32 arithmetic float32 next to
    Python floats).  `oracle.wide_rows()` evaluates them on float64 copies:
    what a DOUBLE build of the device solver is held to.  The two differ by
    float32 rounding of the kernel values, and the default stays the pinned
    one."""
    import networkx as nx
    from graphdot_amd.graph import Graph
    from graphdot_amd.microkernel import (KroneckerDelta, SquareExponential,
                                          TensorProduct)
    from oracle import mgk
    g = nx.cycle_graph(5)
    for v in g.nodes:
        g.nodes[v]['radius'] = [1.0, 1.5, 2.0][v % 3]
    for e in g.edges:
        g.edges[e]['w'] = 1.0
        g.edges[e]['length'] = 0.7 + 0.31 * e[0]
    G = Graph.unify_datatype([Graph.from_networkx(g, weight='w')])
    assert G[0].nodes._data['radius'].dtype == np.float32
    kn = TensorProduct(radius=SquareExponential(0.7454643033504345))
    ke = TensorProduct(length=SquareExponential(0.8810140899648293))
    assert mgk.WIDE_ROWS is False
    narrow = mgk.gram(G, kn, ke, q=0.05)
    with mgk.wide_rows():
        assert mgk.WIDE_ROWS is True
        wide = mgk.gram(G, kn, ke, q=0.05)
    assert mgk.WIDE_ROWS is False
    again = mgk.gram(G, kn, ke, q=0.05)
    assert np.array_equal(narrow, again)
    rel = abs(float(wide[0, 0] / narrow[0, 0]) - 1)
    assert 1e-10 < rel < 1e-6, rel
    # float64 evaluation by hand: the same numbers as wide rows
    r = np.asarray(G[0].nodes._data['radius'], dtype=np.float64)
    V = np.exp(-0.5 * (r[:, None] - r[None, :])**2 / 0.7454643033504345**2)
    s = mgk.PairSide(G[0], wide=True)
    assert np.allclose(mgk.node_table(kn, s, s), V, rtol=1e-15)
