"""Round 6, review item 2a: does the SECOND solve of the value + gradient path
(A y = p1 (x) p2, reference compute_duo: marginalized_kernel.h:492-557) get
cheaper when it is seeded from the Krylov space of the FIRST
(A x = Dx q^2/q0^2)?  Galerkin seed: y0 = sum_k (p_k . b2 / p_k . A p_k) p_k
over the search directions p_k of the first PCG run -- the coefficient's
numerator is the `pp . p_k` the value sum already forms, the denominator the
iteration's own pAp; r2 = b2 - sum_k c_k A p_k comes with it.

CPU only (numpy on the oracle's dense assembly), the sequential stopping rules
of the product (`SEQ`: rTr_0 < tol^2 / 2, rTr_1 < tol^2 - rTr_0, tol =
1e-10 * 2N).  Prints iteration counts of the second solve from zero and from
the seed, per pair sample of the 1000-graph set.
Usage: python scripts/galerkin_seed_experiment.py [--pairs 300] [--float-scalars]"""
import argparse
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases                                      # noqa: E402
from oracle import mgk                            # noqa: E402


def pcg(A, dinv, b, tol2, x0=None, r0=None, seed_rhs=None, cap=None):
    """Jacobi-PCG with the reference's update order.  Returns (x, iterations,
    rTr, seed): `seed` = (y0, r2) for `seed_rhs` accumulated over the search
    directions, or None."""
    N = len(b)
    x = np.zeros(N) if x0 is None else x0.copy()
    r = b.copy() if r0 is None else r0.copy()
    z = dinv * r
    p = z.copy()
    rTz = r @ z
    rTr = r @ r
    y0 = r2 = None
    if seed_rhs is not None:
        y0, r2 = np.zeros(N), seed_rhs.copy()
    k = 0
    cap = cap or 2 * N + 16
    if rTr < tol2:
        return x, 0, rTr, (y0, r2)
    while k < cap and rTz != 0:
        Ap = A @ p
        pAp = p @ Ap
        if pAp == 0:
            break
        if seed_rhs is not None:
            # the residual of the seeded system, not its right-hand side:
            # p_k . r2 = p_k . b2 for A-conjugate directions, and stays so in
            # floating point
            c = (p @ r2) / pAp
            y0 += c * p
            r2 -= c * Ap
        alpha = rTz / pAp
        x += alpha * p
        r -= alpha * Ap
        z = dinv * r
        rTr = r @ r
        rTz_next = r @ z
        k += 1
        if rTr < tol2:
            break
        p = z + (rTz_next / rTz) * p
        rTz = rTz_next
    return x, k, rTr, (y0, r2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=300)
    ap.add_argument('--graphs', type=int, default=1000)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    graphs = cases.config3_graphs(a.graphs)
    knode, kedge, q = cases.config3_kernels()
    rng = np.random.default_rng(a.seed)
    rows = []
    for _ in range(a.pairs):
        i, j = sorted(rng.integers(0, a.graphs, size=2))
        s1, s2 = mgk._side(graphs[i]), mgk._side(graphs[j])
        V = mgk.node_table(knode, s1, s2)
        E = mgk.edge_table(kedge, s1, s2)
        A, Dx = mgk.assemble(s1, s2, V, E, q)
        N = len(Dx)
        dinv = 1.0 / np.diag(A)
        b1 = Dx * 1.0                       # q^2 / q0^2 = 1
        b2 = np.kron(np.full(s1.n, 1.0 / s1.n), np.full(s2.n, 1.0 / s2.n))
        tol2 = (1e-10 * 2 * N) ** 2
        x, k1, rr1, (y0, r2) = pcg(A, dinv, b1, tol2 / 2, seed_rhs=b2)
        y_cold, k2_cold, _, _ = pcg(A, dinv, b2, tol2 - rr1)
        y_seed, k2_seed, _, _ = pcg(A, dinv, b2, tol2 - rr1, x0=y0, r0=r2)
        yref = np.linalg.solve(A, b2)
        rows.append(dict(N=N, k1=k1, k2_cold=k2_cold, k2_seed=k2_seed,
                         seed_resid=float(np.sqrt(r2 @ r2 / (b2 @ b2))),
                         err_cold=float(np.abs(y_cold - yref).max()
                                        / np.abs(yref).max()),
                         err_seed=float(np.abs(y_seed - yref).max()
                                        / np.abs(yref).max())))
    k1 = np.array([r['k1'] for r in rows], float)
    kc = np.array([r['k2_cold'] for r in rows], float)
    ks = np.array([r['k2_seed'] for r in rows], float)
    w = np.array([r['N'] for r in rows], float)     # work ~ N per iteration
    summary = dict(
        pairs=len(rows), mean_k1=k1.mean(), mean_k2_cold=kc.mean(),
        mean_k2_seed=ks.mean(),
        second_solve_iterations_saved=1 - ks.sum() / kc.sum(),
        second_solve_work_saved=1 - (ks * w).sum() / (kc * w).sum(),
        both_solves_work_saved=1 - ((k1 + ks) * w).sum() / ((k1 + kc) * w).sum(),
        mean_seed_residual=float(np.mean([r['seed_resid'] for r in rows])),
        worst_err_cold=max(r['err_cold'] for r in rows),
        worst_err_seed=max(r['err_seed'] for r in rows))
    print(json.dumps(summary, indent=1))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(dict(summary=summary, rows=rows), f, indent=1)


if __name__ == '__main__':
    main()
