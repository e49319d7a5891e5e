"""Marginalized graph kernel on an explicit list of graph pairs.

The API of the reference's ``AltMarginalizedGraphKernel``
(experimental/alterantive_mgk/_kernel.py:11-108): ``kernel(X, ij, lmin=0)``
returns one similarity per index pair instead of a matrix.  On the HIP
backend this is the packed output mode of the solver that the multi-GPU
sharding uses (one value per job, in job order), so nothing is computed or
stored for pairs that were not asked for.  ``eval_gradient`` is an extension.
"""
import numpy as np
from ...graph import Graph
from ._kernel import MarginalizedGraphKernel, _job_t, _type_error


class AltMarginalizedGraphKernel(MarginalizedGraphKernel):

    def __call__(self, X, ij, lmin=0, eval_gradient=False, timing=False):
        """Similarities of the graph pairs ``(X[i], X[j]) for i, j in ij``.

        Returns a vector with one entry per pair (and, with eval_gradient,
        the ``(len(ij), n_active_theta)`` gradient)."""
        backend = self.backend
        if not hasattr(backend, 'prepare'):
            raise TypeError('the pair-list kernel needs the HIP backend')
        pred = Graph.has_unified_types(X)
        if pred is not True:
            raise _type_error(
                pred, 'If the attributes match in name but differ in type, '
                'try `Graph.unify_datatype` as an automatic fix.')
        ij = np.asarray(ij, dtype=np.uint32).reshape(-1, 2)
        if len(ij) and int(ij.max()) >= len(X):
            raise IndexError('pair index beyond the graph list')
        jobs = np.ascontiguousarray(ij).ravel().view(_job_t)
        traits = self.traits(symmetric=False, lmin=lmin,
                             eval_gradient=eval_gradient)
        starts = np.arange(len(X) + 1, dtype=np.uint32)
        plan = backend.prepare(X, self.node_kernel, self.edge_kernel, self.p,
                               self.q, self.eps, self.ftol, self.gtol, jobs,
                               starts, len(X), len(X), self.n_dims, traits,
                               packed=True)
        backend.launch(plan)
        values, grad = backend.collect(plan)
        values = values.astype(self.element_dtype)
        if eval_gradient:
            grad = grad.reshape(len(jobs), self.n_dims)[
                :, np.asarray(self.active_theta_mask)]
            return values, grad.astype(self.element_dtype)
        return values
