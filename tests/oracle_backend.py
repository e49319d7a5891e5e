"""TEST INFRASTRUCTURE: a `Backend` for MarginalizedGraphKernel whose solver
is the CPU oracle (oracle/mgk_oracle.c through `TensorProductBatch`, OpenMP
over the pairs) -- the checker of tests that drive the *callers* of the hot
path (the GPR fit) once with the HIP backend and once with the oracle.
Graph-level evaluations of tensor-product microkernels only.  Nothing under
graphdot_amd/ imports this module."""
import numpy as np
from graphdot_amd.kernel.marginalized._backend import Backend
from oracle import mgk


class OracleBackend(Backend):
    real = np.float64

    def __init__(self, tol=None):
        #: stopping tolerance of value solves (None: the caller's ftol, the
        #: reference's rule sqrt(rTr) < ftol N, marginalized_kernel.h:449)
        self.tol = tol
        self._batch = None
        self.calls = 0

    def __deepcopy__(self, memo):
        # (clones of a kernel share its backend: _backend_cuda.py:63-64)
        return self

    @staticmethod
    def array(a):
        return np.array(a)

    @staticmethod
    def zeros(size, dtype=np.float32):
        return np.zeros(size, dtype=dtype)

    @staticmethod
    def empty(size, dtype=np.float32):
        return np.empty(size, dtype=dtype)

    def __call__(self, graphs, node_kernel, edge_kernel, p, q, eps, ftol,
                 gtol, jobs, starts, gramian, gradient, nX, nY, nJ, traits,
                 timer):
        if traits.nodal is not False or traits.diagonal:
            raise NotImplementedError('graph-level outputs only')
        self.calls += 1
        key = tuple(map(id, graphs))
        if self._batch is None or self._batch[0] != key:
            self._batch = (key, mgk.TensorProductBatch(
                graphs, node_kernel, edge_kernel), list(graphs))
        batch = self._batch[1]
        # (the packed graphs stay; the hyperparameters are this call's)
        _, _, batch.vparam = mgk._tp_spec(node_kernel)
        _, _, batch.eparam = mgk._tp_spec(edge_kernel)
        ji = np.asarray(jobs['i'], dtype=np.int64)
        jj = np.asarray(jobs['j'], dtype=np.int64)
        pv = float(p.p)
        if traits.eval_gradient is True:
            v, g, _ = batch.run_gradient(ji, jj, p=pv, q=q, lmin=traits.lmin,
                                         real='f64', omp=True)
        else:
            v, _ = batch.run(ji, jj, p=pv, q=q, lmin=traits.lmin, real='f64',
                             tol=self.tol or ftol, omp=True)
        a = np.asarray(starts)[ji].astype(np.int64)
        b = np.asarray(starts)[jj].astype(np.int64)
        dst = a + nX * b
        gramian[dst] = v
        if traits.symmetric:
            gramian[b + nX * a] = v
        if traits.eval_gradient is True:
            for k in range(nJ):
                plane = gradient[k * nX * nY:(k + 1) * nX * nY]
                plane[dst] = g[:, k]
                if traits.symmetric:
                    plane[b + nX * a] = g[:, k]
