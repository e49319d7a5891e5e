"""Maximin graph distance on top of the node-wise marginalized graph kernel.

Behaviour of the reference's ``graphdot.metric.maximin.MaxiMin``
(metric/maximin/_maximin.py:11-220 and _backend.cu:40-407): with the nodal
similarities ``k12(i1, i2)`` of a graph pair and the nodal self-similarities
``k1(i1)``, ``k2(i2)``, the kernel-induced node distance is

    d(i1, i2) = sqrt(max(0, 0.9999995 - k12 / sqrt(k1 k2)))     (_backend.cu:24-31,132-134)

and the graph distance the Hausdorff-like

    D = max( max_i1 min_i2 d,  max_i2 min_i1 d )                (_backend.cu:145-166)

with the *hotspot* the node pair that attains it (the largest flat index
``i1 n2 + i2`` among ties, _backend.cu:173-183) and the gradient
``-0.5 d(k12 / sqrt(k1 k2))/dtheta / (D + 1e-4)`` taken at the hotspot
(_backend.cu:136-140,222-250,380-402).

The reference fuses this into its solver kernel, and so does the HIP backend
for graphs its owner-computes solvers cover (``HIPBackend.maximin_distance``:
the reductions run in LDS in the solver's epilogue, the gradient is taken at
the hotspot inside the launch, the nodal Gram matrix is never materialised).
Other graphs / backends take the composition below: nodal matrices from the
solver (``nodal=True`` outputs, finite-difference nodal gradients) and
segmented numpy reductions over the node blocks.
"""
import numpy as np
from ...kernel.marginalized import MarginalizedGraphKernel

_ONE = np.float32(0.9999995)     # cancels round-off so that d(G, G) == 0
_EPS = np.float32(1e-4)          # keeps the gradient finite at d -> 0


def _segments(graphs):
    sizes = np.array([len(g.nodes) for g in graphs], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(sizes)))
    return sizes, starts


class MaxiMin(MarginalizedGraphKernel):
    """Maximin distance between graphs; constructor arguments as for
    :class:`MarginalizedGraphKernel`."""

    def __init__(self, *args, reference_compat=False, **kwargs):
        """reference_compat: form the gradient exactly as the reference's
        kernel does -- its final loop (_backend.cu:380-402) reads k12, and with
        it the distance in the denominator, from the solution buffer *after*
        the finite-difference loop has left the last perturbed solve
        (theta_last e^-eps) there, for every column from q on.  Default
        (False): the unperturbed solution, i.e. the formula the reference's
        comments state.  The two differ by O(eps); the compat form needs the
        fused device path (the host composition never sees the perturbed
        solutions)."""
        kwargs['dtype'] = np.float32
        super().__init__(*args, **kwargs)
        self.reference_compat = bool(reference_compat)

    def __call__(self, X, Y=None, eval_gradient=False, lmin=0,
                 return_hotspot=False, timing=False):
        """Distance matrix ``(len(X), len(Y or X))``; optionally the hotspot
        node indices ``(i1, i2)`` per pair and the gradient w.r.t. the active
        hyperparameters."""
        fused = self._fused(X, Y, eval_gradient, lmin, return_hotspot)
        if fused is not None:
            return fused
        if self.reference_compat and eval_gradient:
            raise NotImplementedError(
                'MaxiMin(reference_compat=True) gradients need the fused '
                'device path (HIP backend, graphs the owner-computes solvers '
                'cover)')
        mgk = super()
        Yl = X if Y is None else Y
        nx_, sx = _segments(X)
        ny_, sy = _segments(Yl)
        if eval_gradient:
            K, dK = mgk.__call__(X, Y, nodal=True, lmin=lmin,
                                 eval_gradient=True, timing=timing)
            k1, dk1 = mgk.diag(X, nodal=True, lmin=lmin, eval_gradient=True)
            k2, dk2 = (k1, dk1) if Y is None else mgk.diag(
                Y, nodal=True, lmin=lmin, eval_gradient=True)
        else:
            K = mgk.__call__(X, Y, nodal=True, lmin=lmin, timing=timing)
            k1 = mgk.diag(X, nodal=True, lmin=lmin)
            k2 = k1 if Y is None else mgk.diag(Y, nodal=True, lmin=lmin)
        K = np.asarray(K, dtype=np.float64)
        k1 = np.asarray(k1, dtype=np.float64)
        k2 = np.asarray(k2, dtype=np.float64)
        rs = 1.0 / np.sqrt(k1[:, None] * k2[None, :])
        d = np.sqrt(np.maximum(0.0, _ONE - K * rs)).astype(np.float32)

        # min over the nodes of the other graph, then max over the own nodes
        d12 = np.maximum.reduceat(
            np.minimum.reduceat(d, sy[:-1], axis=1), sx[:-1], axis=0)
        d21 = np.maximum.reduceat(
            np.minimum.reduceat(d, sx[:-1], axis=0), sy[:-1], axis=1)
        D = np.maximum(d12, d21)
        out = [D.astype(self.element_dtype)]

        if return_hotspot or eval_gradient:
            # largest flat index i1 * n2 + i2 (node numbers within the pair)
            # among the entries equal to the pair's distance
            gx = np.repeat(np.arange(len(X)), nx_)
            gy = np.repeat(np.arange(len(Yl)), ny_)
            l1 = np.arange(sx[-1]) - sx[gx]
            l2 = np.arange(sy[-1]) - sy[gy]
            flat = l1[:, None] * ny_[gy][None, :] + l2[None, :]
            hit = d == D[gx][:, gy]
            flat = np.where(hit, flat, -1)
            hot = np.maximum.reduceat(
                np.maximum.reduceat(flat, sy[:-1], axis=1), sx[:-1], axis=0)
            h1, h2 = hot // ny_[None, :], hot % ny_[None, :]
            if return_hotspot:
                out.append((h1, h2))
        if eval_gradient:
            a = sx[:-1, None] + h1            # global node rows of the hotspots
            b = sy[None, :-1] + h2
            k12 = K[a, b]
            dk12 = np.asarray(dK, dtype=np.float64)[a, b, :]
            ka, kb = k1[a], k2[b]
            dka = np.asarray(dk1, dtype=np.float64)[a, :]
            dkb = np.asarray(dk2, dtype=np.float64)[b, :]
            kk = (ka * kb)[..., None]
            dnorm = dk12 / np.sqrt(kk) - 0.5 * k12[..., None] * kk**-1.5 * (
                dka * kb[..., None] + ka[..., None] * dkb)
            grad = -0.5 * dnorm / (D.astype(np.float64)[..., None] + _EPS)
            # (the starting-probability columns take the same formula on the
            # analytic d k12 / dp, _backend.cu:222-250)
            out.append(grad.astype(self.element_dtype))
        return out[0] if len(out) == 1 else tuple(out)

    def _fused(self, X, Y, eval_gradient, lmin, return_hotspot):
        """The device-fused evaluation, or None if the backend / the graphs
        do not offer it."""
        backend = self.backend
        if not hasattr(backend, 'maximin_distance') or \
                getattr(backend, 'shards_over_ranks', lambda: False)():
            return None
        from ...graph import Graph
        from ...kernel.marginalized._backend_hip import NotOwnerComputes
        graphs = list(X) if Y is None else list(X) + list(Y)
        if Graph.has_unified_types(graphs) is not True:
            return None            # let the composition raise the type error
        nx, ny = len(X), len(X if Y is None else Y)
        if Y is None:
            i, j = np.triu_indices(nx)
        else:
            i, j = np.indices((nx, ny))
            j = j + nx
        job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
        jobs = np.column_stack((i.ravel(), j.ravel())).astype(
            np.uint32).ravel().view(job_t)
        traits = self.traits(symmetric=Y is None, nodal=False, lmin=lmin,
                             eval_gradient=eval_gradient)
        try:
            d, hot, g = backend.maximin_distance(
                graphs, self.node_kernel, self.edge_kernel, self.p, self.q,
                self.eps, self.ftol, self.gtol, jobs, nx, ny, self.n_dims,
                traits, reference_compat=self.reference_compat)
        except NotOwnerComputes:
            return None
        out = [d.reshape(nx, ny, order='F').astype(self.element_dtype)]
        if return_hotspot:
            n = np.array([len(g_.nodes) for g_ in (X if Y is None else Y)])
            hot = hot.reshape(nx, ny, order='F')
            out.append((hot // n[None, :], hot % n[None, :]))
        if eval_gradient:
            g = g.reshape(nx, ny, self.n_dims, order='F')
            out.append(g[:, :, np.asarray(self.active_theta_mask)].astype(
                self.element_dtype))
        return out[0] if len(out) == 1 else tuple(out)

