"""CPU restatement of the maximin graph distance -- TEST INFRASTRUCTURE ONLY
(like mgk.py: nothing under graphdot_amd/ imports this module).

Follows the reference's fused kernel
/root/reference/graphdot/metric/maximin/_backend.cu on the *raw* nodal
solutions ``x`` (n1 x n2) of the pair system -- the solver's output before
post-processing -- of a graph pair and of the perturbed systems of its
finite-difference loop:

* post-processing ``k12 = (x - [lmin] kappa_v q^2/q0^2) p1 p2``      (:119-124)
* node distance ``d = sqrtf(max(0, 0.9999995 - k12 rsqrtf(k1 k2)))``  (:126-128)
* row minima, column minima, their maximum D                        (:140-166)
* hotspot: the largest flat index ``i1 n2 + i2`` with d == D, and for the
  mirrored entry the largest ``i2 n1 + i1``                          (:168-185)
* gradient at the hotspot, columns [p..., q, node theta..., edge theta...]:
  ``-0.5 (dk12 rs - 0.5 k12 rs^3 (dk1 k2 + k1 dk2)) / (d + 1e-4)``     (:130-134)
  - p columns: ``dk12 = (x - corr)(p1 dp2 + p2 dp1)``, finished *before* the
    finite-difference loop, with k12 and d of the unperturbed solve (:222-250)
  - other columns: ``dk12 = (x+[hot] - x-[hot]) / (2 eps theta_j) p1 p2``
    (:252-378), and then the final loop (:380-402) reads k12 -- and d with it --
    from the solution buffer, which at that point holds the LAST perturbed
    solve (minus perturbation of the last hyperparameter).
    `reference_compat=True` restates exactly that; `False` uses the
    unperturbed solution for k12 and d (what the formula in the reference's
    comments says, and this repo's default).

The distance arithmetic is float32 like the reference's (the equality test
``d == D`` that picks the hotspot is a float32 comparison there).
"""
import numpy as np

ONE = np.float32(0.9999995)
EPS = np.float32(1e-4)


def postproc(x, p1, p2, corr=0.0):
    """k12 of every node pair from the raw solution (:119-124).  p1 (n1,),
    p2 (n2,): starting probabilities; corr: kappa_v q^2/q0^2 (n1, n2) for
    lmin = 1, else 0."""
    return (np.asarray(x, dtype=np.float64) - corr) * p1[:, None] * p2[None, :]


def node_distance(k12, k1, k2):
    """float32 matrix of kernel-induced node distances (:126-128)."""
    k12, k1, k2 = (np.asarray(a, dtype=np.float32) for a in (k12, k1, k2))
    rs = np.float32(1) / np.sqrt(k1[:, None] * k2[None, :], dtype=np.float32)
    return np.sqrt(np.maximum(np.float32(0), ONE - k12 * rs),
                   dtype=np.float32)


def pair_distance(d):
    """(D, hotspot flat index i1 n2 + i2, mirrored hotspot i2 n1 + i1) of a
    distance matrix (:140-185)."""
    n1, n2 = d.shape
    D = max(d.min(axis=1).max(), d.min(axis=0).max())
    i1, i2 = np.nonzero(d == D)
    return np.float32(D), int((i1 * n2 + i2).max()), int((i2 * n1 + i1).max())


def normalized_kernel_grad(k12, dk12, k1, dk1, k2, dk2):
    """d(k12 / sqrt(k1 k2)) (:130-134)."""
    kk = k1 * k2
    return dk12 / np.sqrt(kk) - 0.5 * k12 * kk**-1.5 * (dk1 * k2 + k1 * dk2)


def pair_gradient(x0, x_plus, x_minus, denom, p1, p2, dp1, dp2, k1, dk1, k2,
                  dk2, corr=0.0, reference_compat=False):
    """Distance, hotspot and gradient of one pair.

    x0: raw unperturbed solution (n1, n2); x_plus / x_minus: raw solutions of
    the perturbed systems, one pair per finite-difference column in the order
    [q, node theta..., edge theta...]; denom[j] = 2 eps theta_j; dp1 (n_p, n1),
    dp2 (n_p, n2): Jacobians of the starting probabilities; k1 (n1,), k2 (n2,)
    nodal self-similarities with Jacobians dk1 (n1, nJ), dk2 (n2, nJ), nJ =
    n_p + len(x_plus)."""
    n1, n2 = x0.shape
    k12 = postproc(x0, p1, p2, corr)
    d = node_distance(k12, k1, k2)
    D, hot, hot_m = pair_distance(d)
    i1, i2 = hot // n2, hot % n2
    n_p = len(dp1)
    nJ = n_p + len(x_plus)
    grad = np.zeros(nJ)
    c = corr[i1, i2] if np.ndim(corr) else corr
    xh = float(x0[i1, i2]) - c
    kk12, dd = float(k12[i1, i2]), float(D)
    for j in range(n_p):
        dk12 = xh * (p1[i1] * dp2[j][i2] + p2[i2] * dp1[j][i1])
        grad[j] = -0.5 * normalized_kernel_grad(
            kk12, dk12, k1[i1], dk1[i1, j], k2[i2], dk2[i2, j]) / (dd + EPS)
    if reference_compat and len(x_minus):
        # the buffer holds the last perturbed solve when the final loop runs
        kk12 = (float(x_minus[-1][i1, i2]) - c) * p1[i1] * p2[i2]
        dd = float(node_distance(np.array([[kk12]]), k1[i1:i1 + 1],
                                 k2[i2:i2 + 1])[0, 0])
    for t, (xp, xm, den) in enumerate(zip(x_plus, x_minus, denom)):
        j = n_p + t
        dk12 = (float(xp[i1, i2]) - float(xm[i1, i2])) / den * p1[i1] * p2[i2]
        grad[j] = -0.5 * normalized_kernel_grad(
            kk12, dk12, k1[i1], dk1[i1, j], k2[i2], dk2[i2, j]) / (dd + EPS)
    return D, hot, hot_m, grad


def nodal_self(x0, x_plus, x_minus, denom, p, dp, corr=0.0):
    """Nodal self-similarities of one graph and their Jacobian as the
    reference's `diag(nodal=True, eval_gradient=True)` defines them
    (template.cu:226-418): k = diag((x - corr) p p), p columns analytic,
    the others central differences of the raw solutions times p p."""
    n = len(p)
    idx = np.arange(n)
    c = corr[idx, idx] if np.ndim(corr) else corr
    xh = x0[idx, idx] - c
    k = xh * p * p
    cols = [xh * 2 * p * dp[j] for j in range(len(dp))]
    for xp, xm, den in zip(x_plus, x_minus, denom):
        cols.append((xp[idx, idx] - xm[idx, idx]) / den * p * p)
    return k, np.stack(cols, axis=1)
