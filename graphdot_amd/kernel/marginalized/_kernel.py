"""``MarginalizedGraphKernel``: host-side driver of the Gram-matrix hot path.

Restates the reference's ``graphdot/kernel/marginalized/_kernel.py:17-508`` --
job list, output offsets, backend call, F-order result shaping and the
sklearn-style hyperparameter plumbing -- against the HIP backend.  The call
signatures, return shapes, dtype handling, Jacobian column order
``[p..., q, node..., edge...]`` and the 'fixed'-bounds masking are the
reference's, so ``kernel.fix`` / ``model.gaussian_process`` style consumers can
use it unchanged.
"""
import copy
import itertools as it
import numbers
import warnings
from collections import namedtuple
from collections import OrderedDict
import numpy as np
from ...graph import Graph
from ...hip import hostlib
from ...util import Timer
from ...util.iterable import fold_like, flatten, replace
from ...util.pretty_tuple import pretty_tuple
from ._backend_factory import backend_factory
from .starting_probability import StartingProbability, Uniform, Adhoc

_job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])


def _type_error(pred, hint):
    group, first, second = pred
    return TypeError(
        f'The two graphs have mismatching {group} attributes or attribute '
        f'types. {hint}\nFirst graph: {first}\nSecond graph: {second}\n')


class LocalGradient:
    """One rank's share of a pair-sharded symmetric gradient: `dK[p, :]` (a
    device tensor, all `n_dims` columns) belongs to the pair `(i[p], j[p])`,
    i <= j; the ranks of `group` hold every unordered pair exactly once."""

    def __init__(self, dK, i, j, group=None):
        self.dK, self.i, self.j, self.group = dK, i, j, group
        #: indices of the columns the consumer wants (None: all)
        self.columns = None


class PendingLocalGradient(LocalGradient):
    """A `LocalGradient` whose solvers may still be running on their own
    streams (`device_gram(..., local_gradient='overlapped')`): what the caller
    enqueues on the null stream meanwhile overlaps them; reading `dK` first
    orders the null stream behind the solvers."""

    def __init__(self, step, i, j, group=None):
        self._step, self.i, self.j, self.group = step, i, j, group
        self.columns = None

    @property
    def dK(self):
        self._step.join()
        return self._step.local_gradient


class MarginalizedGraphKernel:
    """Random-walk graph kernel of Kashima, Tsuda & Inokuchi (ICML 2003) in
    the linear-system form of Tang & de Jong (J. Chem. Phys. 150, 044107).

    Parameters
    ----------
    node_kernel, edge_kernel: microkernels
        Similarity between individual nodes / edges.
    p: positive number (default 1.0), :py:class:`StartingProbability`, or a
       ``(callable, "C++ expression")`` pair
        Starting probability of the random walk.
    q: float in (0, 1)
        Stopping probability.
    q_bounds: (float, float)
    eps: float
        Finite-difference step (log-scale) for nodal gradients.
    ftol, gtol: float
        Solver tolerances for the value and for finite-difference re-solves.
    dtype: numpy dtype of the returned matrices
    backend: 'auto' | 'hip' | :py:class:`Backend` instance
    """
    trait_t = namedtuple(
        'Traits', 'diagonal, symmetric, nodal, lmin, eval_gradient')

    @classmethod
    def traits(cls, diagonal=False, symmetric=False, nodal=False, lmin=0,
               eval_gradient=False):
        return cls.trait_t(diagonal, symmetric, nodal, lmin, eval_gradient)

    def __init__(self, node_kernel, edge_kernel, p=1.0, q=0.01,
                 q_bounds=(1e-4, 1 - 1e-4), eps=1e-2, ftol=1e-8, gtol=1e-6,
                 dtype=float, backend='auto'):
        self.node_kernel = node_kernel
        self.edge_kernel = edge_kernel
        self.p = self._get_starting_probability(p)
        self.q = q
        self.q_bounds = q_bounds
        self.eps = eps
        self.ftol = ftol
        self.gtol = gtol
        self.element_dtype = dtype
        self.backend = backend_factory(backend)

        lo, hi = self.node_kernel.minmax
        if lo <= 0 or hi > 1:
            warnings.warn(
                'Node kernel value range should be within (0, 1], '
                f'got {self.node_kernel.minmax} for {self.node_kernel}. '
                'This will not be allowed in a future version. '
                'Consider adding a small constant or using the `.normalized` '
                'attribute of the kernel.', DeprecationWarning)
        lo, hi = self.edge_kernel.minmax
        if lo < 0 or hi > 1:
            warnings.warn(
                'Edge kernel value range must be within [0, 1], '
                f'got {self.edge_kernel.minmax} for {self.edge_kernel}. '
                'This will not be allowed in a future version. '
                'Consider adding a small constant or using the `.normalized` '
                'attribute of the kernel.', DeprecationWarning)

    @staticmethod
    def _get_starting_probability(p):
        if isinstance(p, StartingProbability):
            return p
        if isinstance(p, tuple) and len(p) == 2:
            f, expr = p
            if callable(f) and isinstance(expr, str):
                return Adhoc(f, expr)
            raise ValueError('An ad hoc starting probability must be '
                             'specified as an (callable, C++ expression) '
                             'pair.')
        if isinstance(p, numbers.Number):
            if p > 0:
                return Uniform(p)
            raise ValueError(f'Starting probability {p} < 0.')
        raise ValueError(f'Unknown starting probability: {p}')

    # ------------------------------------------------------------------ Gram
    _jobs_cache = OrderedDict()     # (nx, ny) -> read-only job list (shared)

    def _pairwise_jobs(self, nx, ny=None):
        """The job list of the reference (_kernel.py:172-182): the upper
        triangle including the diagonal for a symmetric matrix, all (i, nx+j)
        otherwise.  Lists are kept (read-only) so that the backend can
        recognise a repeated evaluation by the identity of the array."""
        cache = MarginalizedGraphKernel._jobs_cache
        key = (nx, ny)
        if key in cache:
            cache.move_to_end(key)
            return cache[key]
        # (natively: np.triu_indices builds an nx x nx mask, 3.5 ms for 1000
        # graphs on the first call.  Only a backend on the native host path
        # needs libgdhost: any other backend of the plugin seam, and
        # HIPBackend(native=False), take the numpy form)
        if getattr(self.backend, 'native', False):
            jobs = hostlib.pairwise_jobs(nx, ny, _job_t)
        else:
            if ny is None:
                i, j = np.triu_indices(nx)
            else:
                i, j = np.indices((nx, ny), dtype=np.uint32)
                j = j + nx
            jobs = np.column_stack((i.ravel(), j.ravel())).astype(
                np.uint32).ravel().view(_job_t)
        if type(self.backend.array(jobs[:0])) is not np.ndarray:
            jobs = self.backend.array(jobs)     # (a backend with its own arrays)
        if isinstance(jobs, np.ndarray):
            jobs.flags.writeable = False
            cache[key] = jobs
            while len(cache) > 8:
                cache.popitem(last=False)
        return jobs

    def __call__(self, X, Y=None, eval_gradient=False, nodal=False, lmin=0,
                 timing=False):
        """Pairwise similarity matrix.

        Parameters
        ----------
        X: list of N graphs (same node and edge attributes)
        Y: None or list of M graphs
        eval_gradient: bool
            Also return the gradient w.r.t. the hyperparameters.
        nodal: bool
            Node-wise instead of graph-wise similarities.
        lmin: 0 or 1
            Number of leading steps of each walk excluded from the similarity.

        Returns
        -------
        K: ndarray (N, N) if Y is None else (N, M)   [node counts if nodal]
        dK: ndarray (..., n_active_theta), only if eval_gradient
        """
        timer = Timer()
        backend = self.backend
        traits = self.traits(symmetric=Y is None, nodal=nodal, lmin=lmin,
                             eval_gradient=eval_gradient)

        all_graphs = list(it.chain(X, Y)) if Y is not None else X
        pred = Graph.has_unified_types(all_graphs)
        if pred is not True:
            raise _type_error(
                pred, 'If the attributes match in name but differ in type, '
                'try `Graph.unify_datatype` as an automatic fix.')

        timer.tic('generating jobs')
        nx = len(X)
        jobs = self._pairwise_jobs(nx, None if traits.symmetric else len(Y))
        timer.toc('generating jobs')

        timer.tic('creating output buffer')
        if traits.symmetric:
            starts = backend.zeros(nx + 1, dtype=np.uint32)
            if traits.nodal is True:
                sizes = np.array([len(g.nodes) for g in X], dtype=np.uint32)
                np.cumsum(sizes, out=starts[1:])
                output_shape = (int(starts[-1]),) * 2
            else:
                starts[:] = np.arange(nx + 1)
                output_shape = (nx, nx)
        else:
            ny = len(Y)
            starts = backend.zeros(nx + ny + 1, dtype=np.uint32)
            if traits.nodal is True:
                sizes = np.array([len(g.nodes) for g in all_graphs],
                                 dtype=np.uint32)
                np.cumsum(sizes, out=starts[1:])
                n_nodes_X = int(starts[nx])
                starts[nx:] -= np.uint32(n_nodes_X)
                output_shape = (n_nodes_X, int(starts[-1]))
            else:
                starts[:nx] = np.arange(nx)
                starts[nx:] = np.arange(ny + 1)
                output_shape = (nx, ny)
        n_out = int(np.prod(output_shape))
        # fp32 like the reference (_kernel.py:212-216) unless the backend
        # computes in double precision
        real = getattr(backend, 'real', np.float32)
        gramian = backend.empty(n_out, real)
        gradient = (backend.empty(self.n_dims * n_out, real)
                    if traits.eval_gradient is True else None)
        timer.toc('creating output buffer')

        timer.tic('calling GPU kernel (overall)')
        # (an empty X or Y: nothing to solve, empty outputs of the right
        # shape instead of a zero-size launch)
        if n_out > 0:
            backend(
                np.concatenate((X, Y)) if Y is not None else X,
                self.node_kernel, self.edge_kernel, self.p, self.q,
                self.eps, self.ftol, self.gtol,
                jobs, starts, gramian, gradient,
                output_shape[0], output_shape[1], self.n_dims,
                traits, timer,
            )
        timer.toc('calling GPU kernel (overall)')

        timer.tic('collecting result')
        gramian = gramian.reshape(*output_shape, order='F')
        if gradient is not None:
            gradient = gradient.reshape(
                (*output_shape, self.n_dims), order='F')
            mask = np.asarray(self.active_theta_mask)
            if not mask.all():      # (a copy; skipped when nothing is fixed)
                gradient = gradient[:, :, mask]
        timer.toc('collecting result')

        if timing:
            timer.report(unit='ms')
        timer.reset()

        if traits.eval_gradient is True:
            return (gramian.astype(self.element_dtype, copy=False),
                    gradient.astype(self.element_dtype, copy=False))
        return gramian.astype(self.element_dtype, copy=False)

    def device_gram(self, X, eval_gradient=False, lmin=0,
                    local_gradient=False):
        """The symmetric Gram matrix of `X` (and its gradient) left in device
        memory: the HIP backend's output buffers as zero-copy views
        (`graphdot_amd.hip.runtime.DeviceArray`; ``torch.as_tensor(view,
        device='cuda')`` adopts them), in the backend's arithmetic, column-
        major like the reference's outputs.  The gradient has all `n_dims`
        columns; `active_theta_mask` is the caller's to apply.  The views are
        valid until the next evaluation on this backend.  For consumers that
        continue on the GPU (model.gaussian_process): saves the download, the
        float64 conversion on the host and the upload of ``n^2 (1 + n_dims)``
        numbers per call.

        `local_gradient=True` (with `eval_gradient`): under a backend that
        shards the pairs over ranks the gradient is NOT all-gathered; the
        second return value is then a `LocalGradient` -- the gradient entries
        of this rank's pairs, ``dK[p, :]`` for pair ``(i[p], j[p])``, every
        unordered pair of the symmetric matrix on exactly one rank -- for a
        consumer that reduces over pairs and all-reduces the result
        (model.gaussian_process: ``sum_ij W_ij dK_ij``).  Any other backend
        returns the full planes as before.  `local_gradient='overlapped'`:
        where the backend finds it worthwhile (`overlaps_dense_algebra`: from
        four ranks up) the matrix comes from a value step of its own and the
        value + gradient solvers are then enqueued DETACHED -- the second
        return value is a `PendingLocalGradient`, and whatever the caller
        enqueues on the null stream before reading its `dK` (a Cholesky
        factorisation) runs beside the gradient solves."""
        from ...hip.runtime import DeviceArray
        backend = self.backend
        if not hasattr(backend, 'prepare'):
            raise TypeError('device_gram needs the HIP backend')
        pred = Graph.has_unified_types(X)
        if pred is not True:
            raise _type_error(
                pred, 'If the attributes match in name but differ in type, '
                'try `Graph.unify_datatype` as an automatic fix.')
        nx = len(X)
        traits = self.traits(symmetric=True, lmin=lmin,
                             eval_gradient=eval_gradient)
        real = np.dtype(backend.real)
        if getattr(backend, 'shards_over_ranks', lambda: False)():
            # pair-sharded over the ranks: the all-gathered, reassembled
            # matrix (and gradient planes) of `ShardedStep` stay on this
            # rank's device -- every rank holds the full result, like the
            # single-GPU path, and nothing goes through host memory
            args = (X, self.node_kernel, self.edge_kernel, self.p, self.q,
                    self.eps, self.ftol, self.gtol, self._pairwise_jobs(nx),
                    np.arange(nx + 1, dtype=np.uint32), nx, nx, self.n_dims)
            if eval_gradient and local_gradient == 'overlapped' \
                    and backend.overlaps_dense_algebra():
                vstep = backend.sharded_step(*args, traits._replace(
                    eval_gradient=False))
                K = DeviceArray.fortran(vstep.result.data_ptr(), (nx, nx),
                                        real, owner=vstep)
                gstep = backend.sharded_step(*args, traits,
                                             gather_gradient=False,
                                             solvers_only=True)
                i, j = gstep.local_index
                return K, PendingLocalGradient(gstep, i, j, gstep.group)
            step = backend.sharded_step(
                *args, traits, gather_gradient=not (eval_gradient
                                                    and local_gradient))
            base = step.result.data_ptr()
            K = DeviceArray.fortran(base, (nx, nx), real, owner=step)
            if not eval_gradient:
                return K
            if local_gradient:
                i, j = step.local_index
                return K, LocalGradient(step.local_gradient, i, j, step.group)
            dK = DeviceArray.fortran(base + nx * nx * real.itemsize,
                                     (nx, nx, self.n_dims), real, owner=step)
            return K, dK
        plan = backend.prepare(
            X, self.node_kernel, self.edge_kernel, self.p, self.q, self.eps,
            self.ftol, self.gtol, self._pairwise_jobs(nx),
            np.arange(nx + 1, dtype=np.uint32), nx, nx, self.n_dims, traits)
        backend.launch(plan)
        backend.synchronize()
        K = DeviceArray.fortran(plan.buffers['gramian'].ptr, (nx, nx), real,
                                owner=plan)
        if not eval_gradient:
            return K
        dK = DeviceArray.fortran(plan.buffers['gradient'].ptr,
                                 (nx, nx, self.n_dims), real, owner=plan)
        return K, dK

    # ------------------------------------------------------------------ diag
    def diag(self, X, eval_gradient=False, nodal=False, lmin=0,
             active_theta_only=True, timing=False):
        """Self-similarities of the graphs in X.

        nodal=False: vector of graph self-similarities; nodal=True: vector of
        all nodal self-similarities; nodal='block': list of per-graph
        node-by-node similarity matrices.  With eval_gradient also returns
        the gradient (columns restricted to the active hyperparameters unless
        ``active_theta_only=False``).
        """
        timer = Timer()
        backend = self.backend
        traits = self.traits(diagonal=True, nodal=nodal, lmin=lmin,
                             eval_gradient=eval_gradient)

        pred = Graph.has_unified_types(X)
        if pred is not True:
            raise _type_error(
                pred, 'If the attribute names do match, then try to unify '
                'data types automatically with `Graph.unify_datatype`.')

        timer.tic('generating jobs')
        i = np.arange(len(X), dtype=np.uint32)
        jobs = backend.array(np.column_stack((i, i)).ravel().view(_job_t))
        timer.toc('generating jobs')

        timer.tic('creating output buffer')
        starts = backend.zeros(len(X) + 1, dtype=np.uint32)
        if nodal is True:
            sizes = np.array([len(g.nodes) for g in X], dtype=np.uint32)
            np.cumsum(sizes, out=starts[1:])
        elif nodal is False:
            starts[:] = np.arange(len(X) + 1)
        elif nodal == 'block':
            sizes = np.array([len(g.nodes) for g in X], dtype=np.uint32)
            np.cumsum(sizes**2, out=starts[1:])
        else:
            raise ValueError("Invalid 'nodal' option '%s'" % nodal)
        output_length = int(starts[-1])
        real = getattr(backend, 'real', np.float32)
        gramian = backend.empty(output_length, real)
        gradient = (backend.empty(self.n_dims * output_length, real)
                    if traits.eval_gradient is True else None)
        timer.toc('creating output buffer')

        timer.tic('calling GPU kernel (overall)')
        backend(
            X, self.node_kernel, self.edge_kernel, self.p, self.q,
            self.eps, self.ftol, self.gtol,
            jobs, starts, gramian, gradient,
            output_length, 1, self.n_dims,
            traits, timer,
        )
        timer.toc('calling GPU kernel (overall)')

        timer.tic('collecting result')
        if gradient is not None:
            gradient = gradient.reshape((output_length, self.n_dims),
                                        order='F')
            if active_theta_only:
                gradient = gradient[:, self.active_theta_mask]
        if nodal == 'block':
            retval = [gramian[s:s + n**2].reshape(n, n)
                      for s, n in zip(starts[:-1], sizes)]
        elif traits.eval_gradient is True:
            retval = (gramian.astype(self.element_dtype),
                      gradient.astype(self.element_dtype))
        else:
            retval = gramian.astype(self.element_dtype)
        timer.toc('collecting result')

        if timing:
            timer.report(unit='ms')
        timer.reset()
        return retval

    # ----------------------------------------- scikit-learn interoperability
    def is_stationary(self):
        return False

    @property
    def requires_vector_input(self):
        return False

    _hyper_fields = ['starting_probability', 'stopping_probability',
                     'node_kernel', 'edge_kernel']

    @property
    def hyperparameters(self):
        """Hierarchical view of all hyperparameters."""
        return pretty_tuple('MarginalizedGraphKernel', self._hyper_fields)(
            self.p.theta, self.q, self.node_kernel.theta,
            self.edge_kernel.theta)

    @property
    def flat_hyperparameters(self):
        return np.fromiter(flatten(self.hyperparameters), float)

    @property
    def hyperparameter_bounds(self):
        return pretty_tuple('GraphKernelHyperparameterBounds',
                            self._hyper_fields)(
            self.p.bounds, self.q_bounds, self.node_kernel.bounds,
            self.edge_kernel.bounds)

    @property
    def n_dims(self):
        """Number of hyperparameters, fixed ones included."""
        return len(self.flat_hyperparameters)

    def _flat_bounds(self):
        """(n_dims, 2) array of bounds with 'fixed' -> (nan, nan)."""
        pairs = replace(flatten(self.hyperparameter_bounds), 'fixed',
                        (np.nan, np.nan))
        return np.fromiter(flatten(pairs), dtype=float).reshape(-1, 2)

    @property
    def active_theta_mask(self):
        lower, upper = self._flat_bounds().T
        return ~(np.isnan(lower) | np.isnan(upper) | (lower == upper))

    @property
    def theta(self):
        """log of the non-fixed hyperparameters, flattened."""
        return np.log(self.flat_hyperparameters[self.active_theta_mask])

    @theta.setter
    def theta(self, value):
        hypers = np.log(self.flat_hyperparameters)
        hypers[self.active_theta_mask] = value
        (self.p.theta, self.q, self.node_kernel.theta,
         self.edge_kernel.theta) = fold_like(np.exp(hypers),
                                             self.hyperparameters)

    @property
    def bounds(self):
        """log of the (n_active, 2) bounds of the non-fixed
        hyperparameters."""
        return np.log(self._flat_bounds()[self.active_theta_mask, :])

    def clone_with_theta(self, theta):
        clone = copy.deepcopy(self)
        clone.theta = theta
        return clone
