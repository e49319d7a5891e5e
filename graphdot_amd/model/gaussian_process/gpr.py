"""Gaussian process regression on top of the kernel protocol.

The caller on the other side of the hot path (SURVEY.md section 8f rank 3,
BASELINE.json configuration 5): the behaviour of the reference's
``graphdot.model.gaussian_process.GaussianProcessRegressor``
(model/gaussian_process/gpr.py:9-415 and base.py:14-189) -- same constructor,
``fit / fit_loocv / predict / predict_loocv / log_marginal_likelihood /
squared_loocv_error / save / load``, same conventions:

* the likelihood objective is ``y^T K^-1 y + log|K|`` (twice the negative log
  likelihood without the constant, gpr.py:287-298) and its gradient is taken
  w.r.t. the log-scale hyperparameters,
  ``(tr(K^-1 dK_k) - a^T dK_k a) exp(theta_k)`` with ``a = K^-1 y``;
* ``alpha`` regularises the diagonal additively or multiplicatively
  (base.py:64-73); targets that are None / NaN are masked out (base.py:45-62);
* a singular kernel matrix falls back to a clamped spectral pseudo-inverse
  with cutoff ``beta`` (base.py:108-127).

What is new: the dense algebra on the ``n x n`` kernel matrix and the
``n x n x n_theta`` gradient (Cholesky, solves, the two contractions of the
gradient) runs in float64 through torch on the GPU the kernel matrix came
from, so that one likelihood step of configuration 5 is kernel + a few ms;
without a GPU the same code runs on torch's CPU backend (tests).
"""
import os
import pickle
import time
import warnings
import numpy as np
from scipy.optimize import minimize


def _torch():
    import torch
    return torch


def _contract_planes(W, dK):
    """``sum_ij W[i, j] dK[i, j, k]`` for a *symmetric* W.  The kernel hands
    its gradient over plane by plane (element (i, j, k) at i + n j + n^2 k,
    reference _kernel.py:249-256): then the contraction is one matrix-vector
    product over contiguous rows, 0.03 ms for n = 1000 and six planes on an
    MI355X against 0.34 ms for multiply + reduce over the broadcast product
    (scripts/time_gpr_dense.py).  Any other layout: multiply + reduce (the
    einsum / GEMV form on a hyperparameter-fastest layout took 107 ms on
    rocBLAS)."""
    n0, n1, nt = dK.shape
    planes = dK.permute(2, 1, 0)                # [k][j][i]: W_ji = W_ij
    if planes.is_contiguous():
        return planes.reshape(nt, n0 * n1) @ W.reshape(n0 * n1)
    if dK.permute(2, 0, 1).is_contiguous():     # [k][i][j]
        return dK.permute(2, 0, 1).reshape(nt, n0 * n1) @ \
            W.contiguous().reshape(n0 * n1)
    return (W.unsqueeze(-1) * dK).sum((0, 1))


def _contract_local(W, lg, keep=None):
    """The same contraction from ONE RANK'S share of a pair-sharded
    symmetric gradient (`LocalGradient` of the marginalized kernel: the rows
    `dK[p, :]` of the pairs `(i[p], j[p])`, i <= j, every unordered pair on
    exactly one rank): ``sum_p m_p W[i_p, j_p] dK[p, :]`` with m = 2 off the
    diagonal, then ONE all-reduce of n_theta numbers over the ranks -- instead
    of an all-gather and a reassembly of n_theta planes of n^2 numbers
    followed by the same contraction on every rank.  `keep`: indices of the
    rows / columns of the full matrix that W covers (masked targets)."""
    torch = _torch()
    import torch.distributed as dist
    i = torch.as_tensor(lg.i, device=W.device)
    j = torch.as_tensor(lg.j, device=W.device)
    if keep is not None:
        # W lives on the kept rows: scatter it into the full index space
        n = int(max(int(i.max()) if len(i) else 0,
                    int(j.max()) if len(j) else 0, int(keep.max())) + 1)
        full = torch.zeros((n, n), dtype=W.dtype, device=W.device)
        full[keep[:, None], keep[None, :]] = W
        W = full
    w = W[i, j] * torch.where(i == j, 1.0, 2.0).to(W.dtype)
    rows = lg.dK          # (a pending gradient: the null stream joins the solvers)
    if lg.columns is not None:
        rows = rows.index_select(1, torch.as_tensor(lg.columns,
                                                    device=W.device))
    # (multiply + reduce: as a matrix-vector product of a (pairs x n_theta)
    # matrix this is the skinny GEMV that takes rocBLAS 0.1 us per ROW --
    # 37 ms for the 250 000 pairs of a rank of two, measured)
    d = (rows.to(W.dtype) * w.unsqueeze(1)).sum(dim=0)
    if dist.is_available() and dist.is_initialized() \
            and dist.get_world_size(lg.group) > 1:
        from ...kernel.marginalized._sharded import cuda_collective
        if cuda_collective(lg.group):
            dist.all_reduce(d, group=lg.group)
        else:
            h = d.cpu()
            dist.all_reduce(h, group=lg.group)
            d = h.to(W.device)
    return d


class _Dense:
    """float64 dense algebra on one device."""

    def __init__(self, device='auto'):
        torch = _torch()
        if device == 'auto':
            device = 'cuda' if torch.cuda.is_available() else 'cpu'
            if device == 'cpu' and torch.cuda.device_count() > 0:
                warnings.warn(
                    'a GPU is present but torch cannot use it (its HIP '
                    'runtime was initialised after libgdhip: import '
                    'graphdot_amd.model.gaussian_process, or torch, before '
                    'the first kernel evaluation); dense algebra runs on '
                    'the CPU')
        self.device = torch.device(device)
        #: Cholesky factorisations on the GPU by the kernels of potrf.hip
        #: (GD_NATIVE_CHOLESKY=0: torch.linalg.cholesky_ex)
        self.native_cholesky = os.environ.get('GD_NATIVE_CHOLESKY', '1') != '0'

    def tensor(self, a):
        """float64 device tensor of `a`.  A column-major array (the kernel
        returns its gradient that way: reference _kernel.py:249-256) is
        uploaded as it lies in memory and permuted on the device instead of
        being transposed on the host first."""
        torch = _torch()
        a = np.asarray(a, dtype=np.float64)
        if a.ndim > 1 and a.flags.f_contiguous and not a.flags.c_contiguous:
            t = torch.from_numpy(a.T).to(self.device)
            return t.permute(*reversed(range(a.ndim)))
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def native(self, K):
        """Does `K` take the one-launch factor-and-invert of potrf.hip?"""
        return bool(K.is_cuda and K.dtype == _torch().float64
                    and self.native_cholesky)

    def factor(self, K, rcond, try_native=True):
        """Returns (Kinv as a dense tensor, log-determinant).  Cholesky
        first; a clamped pseudo-inverse if the matrix is not positive
        definite (the reference clamps small eigenvalues to the cutoff:
        base.py:126-127, linalg/spectral.py).  `try_native=False`: the
        caller has already seen the native factorisation fail on `K`."""
        torch = _torch()
        if self.native(K) and not try_native:
            pass
        elif self.native(K):
            # factor and inverse in ONE data-flow launch of potrf.hip (round
            # 6: 31 launches + a triangular solve + X^T X before, 1.24 ms at
            # n = 1000), then ONE small download: the launch's status word
            # and the log-determinant shares of the diagonal blocks -- the
            # only host synchronisation of the factorisation (a pivot that
            # is not positive makes the sum of logarithms NaN)
            from ._potrf import factor_inverse, read_head, FactorisationError
            Kinv, head, nb = factor_inverse(K)
            completed, logdet = read_head(head, nb)
            if not completed:
                raise FactorisationError(
                    'spd_factor_invert_f64 gave up waiting for a tile')
            logdet *= 2.0
            if np.isfinite(logdet):
                return Kinv, logdet
        else:
            L, info = torch.linalg.cholesky_ex(K)
            if int(info) == 0:
                Kinv = torch.cholesky_inverse(L)
                logdet = 2.0 * torch.log(torch.diagonal(L)).sum()
                return Kinv, float(logdet)
        warnings.warn('Kernel matrix singular, falling back to pseudoinverse')
        w, V = torch.linalg.eigh(0.5 * (K + K.T))
        if not bool(torch.isfinite(w).all()):
            raise np.linalg.LinAlgError(
                'The kernel matrix is likely corrupted with NaNs and Infs '
                'because a pseudoinverse could not be computed.')
        cut = rcond * float(w.abs().max())
        w = torch.clamp(w, min=cut)
        Kinv = (V / w) @ V.T
        return Kinv, float(torch.log(w).sum())


class GaussianProcessRegressor:
    """Gaussian process regression.

    Parameters
    ----------
    kernel: kernel instance (``kernel(X, Y=None, eval_gradient=False)``,
        ``.diag(X)``, ``.theta`` log-scale and settable, ``.bounds``,
        ``.clone_with_theta``), e.g. ``MarginalizedGraphKernel``.
    alpha: float > 0
        Regularisation of the diagonal of the kernel matrix.
    beta: float > 0
        Relative eigenvalue cutoff of the pseudo-inverse fallback.
    optimizer: str, True, None or callable
        Method for ``scipy.optimize.minimize``; True means L-BFGS-B; None
        disables hyperparameter optimisation in ``fit``.
    normalize_y: bool
        Standardise the targets for fitting, undo for predictions.
    regularization: '+', 'additive', '*' or 'multiplicative'
    kernel_options: dict
        Extra keyword arguments for every kernel evaluation.
    device: 'auto', 'cuda', 'cpu'
        Where the dense algebra runs.
    """

    def __init__(self, kernel, alpha=1e-8, beta=1e-8, optimizer=None,
                 normalize_y=False, regularization='+', kernel_options={},
                 device='auto'):
        self.kernel = kernel
        self.alpha = alpha
        self.beta = beta
        self.optimizer = 'L-BFGS-B' if optimizer is True else optimizer
        self.normalize_y = normalize_y
        self.regularization = regularization
        self.kernel_options = dict(kernel_options)
        self.device = device

    # -- data ---------------------------------------------------------------
    @property
    def X(self):
        try:
            return self._X
        except AttributeError:
            raise AttributeError(
                'Training data does not exist. Please provide using fit().')

    @X.setter
    def X(self, X):
        self._X = np.asarray(X)

    @property
    def y(self):
        try:
            return self._y * self._ystd + self._ymean
        except AttributeError:
            raise AttributeError(
                'Training data does not exist. Please provide using fit().')

    @staticmethod
    def mask(iterable):
        """(mask of usable targets, the usable targets as float64)."""
        values = list(iterable)
        mask = np.array([v is not None and bool(np.isfinite(v))
                         for v in values], dtype=bool)
        masked = np.array([float(v) for v, m in zip(values, mask) if m],
                          dtype=np.float64)
        return mask, masked

    @y.setter
    def y(self, y):
        self._y_mask, y_masked = self.mask(y)
        if self.normalize_y is True:
            self._ymean, self._ystd = y_masked.mean(), y_masked.std()
            self._y = (y_masked - self._ymean) / self._ystd
        else:
            self._ymean, self._ystd = 0, 1
            self._y = y_masked

    # -- kernel matrices --------------------------------------------------------
    def _regularize(self, K, alpha):
        if self.regularization in ('+', 'additive'):
            return K + alpha
        if self.regularization in ('*', 'multiplicative'):
            return K * (1 + alpha)
        raise RuntimeError(
            f'Unknown regularization method {self.regularization}.')

    def _gramian(self, alpha, X, Y=None, kernel=None, jac=False, diag=False):
        kernel = kernel or self.kernel
        opts = self.kernel_options
        if Y is not None:
            if diag is True:
                raise ValueError(
                    'Diagonal Gramian does not exist between two sets.')
            return kernel(X, Y, eval_gradient=True, **opts) if jac \
                else kernel(X, Y, **opts)
        if diag is True:
            return self._regularize(kernel.diag(X, **opts), alpha)
        if jac is True:
            K, J = kernel(X, eval_gradient=True, **opts)
        else:
            K, J = kernel(X, **opts), None
        K = np.array(K, dtype=np.float64)
        step = len(K) + 1
        K.flat[::step] = self._regularize(K.flat[::step], alpha)
        return (K, J) if jac is True else K

    def _dense(self):
        if not isinstance(getattr(self, '_la', None), _Dense) \
                or self._la_device != self.device:
            self._la, self._la_device = _Dense(self.device), self.device
        return self._la

    # -- fitting -----------------------------------------------------------------
    def fit(self, X, y, loss='likelihood', tol=1e-5, repeat=1,
            theta_jitter=1.0, verbose=False):
        """Train: optionally optimise the hyperparameters against `loss`
        ('likelihood' or 'loocv'), then factor the kernel matrix."""
        self.X = X
        self.y = y
        if self.optimizer:
            if loss == 'likelihood':
                objective = self.log_marginal_likelihood
            elif loss == 'loocv':
                objective = self.squared_loocv_error
            else:
                raise RuntimeError(f'Unknown loss function: {loss}.')
            x0 = np.array(self.kernel.theta, dtype=float)
            starts = [x0] + [x0 + theta_jitter * np.random.randn(len(x0))
                             for _ in range(repeat - 1)]
            best = None
            for x in starts:
                res = minimize(
                    fun=lambda t: objective(t, eval_gradient=True,
                                            clone_kernel=False,
                                            verbose=verbose),
                    method=self.optimizer, x0=x, bounds=self.kernel.bounds,
                    jac=True, tol=tol)
                if best is None or (res.success and res.fun < best.fun):
                    best = res
            if verbose:
                print(f'Optimization result:\n{best}')
            if not best.success:
                raise RuntimeError(
                    f'Training using the {loss} loss did not converge, got:\n'
                    f'{best}')
            self.kernel.theta = best.x
            #: the optimiser's report (scipy OptimizeResult: nit, nfev, fun)
            self.optimization_result = best
        la = self._dense()
        K = self._gramian(self.alpha, self._X)
        self.K = K = K[self._y_mask, :][:, self._y_mask]
        Kinv, _ = la.factor(la.tensor(K), self.beta)
        self.Kinv = Kinv.cpu().numpy()
        self.Ky = self.Kinv @ self._y
        return self

    def fit_loocv(self, X, y, **options):
        return self.fit(X, y, loss='loocv', **options)

    # -- prediction --------------------------------------------------------------
    def predict(self, Z, return_std=False, return_cov=False):
        if not hasattr(self, 'Kinv'):
            raise RuntimeError('Model not trained.')
        Ks = np.asarray(self._gramian(None, Z, self._X),
                        dtype=np.float64)[:, self._y_mask]
        ymean = (Ks @ self.Ky) * self._ystd + self._ymean
        if return_std is True:
            Kss = self._gramian(self.alpha, Z, diag=True)
            var = Kss - np.einsum('ij,jk,ik->i', Ks, self.Kinv, Ks)
            return ymean, np.sqrt(np.maximum(0, var)) * self._ystd
        if return_cov is True:
            Kss = self._gramian(self.alpha, Z)
            cov = np.maximum(0, Kss - Ks @ (self.Kinv @ Ks.T))
            return ymean, cov * self._ystd**2
        return ymean

    def predict_loocv(self, Z, z, return_std=False):
        z_mask, z_masked = self.mask(z)
        if self.normalize_y is True:
            z_mean, z_std = z_masked.mean(), z_masked.std()
            z = (z_masked - z_mean) / z_std
        else:
            z_mean, z_std = 0, 1
            z = z_masked
        la = self._dense()
        K = self._gramian(self.alpha, Z)[z_mask, :][:, z_mask]
        Kinv = la.factor(la.tensor(K), self.beta)[0].cpu().numpy()
        d = Kinv.diagonal()
        ymean = (z - Kinv @ z / d) * z_std + z_mean
        if return_std is True:
            return ymean, np.sqrt(1 / np.maximum(d, 1e-14)) * z_std
        return ymean

    # -- objectives ----------------------------------------------------------------
    def _objective_inputs(self, theta, X, y, eval_gradient, clone_kernel,
                          local_gradient=False):
        theta = np.array(theta if theta is not None else self.kernel.theta,
                         dtype=float)
        X = X if X is not None else self._X
        if y is not None:
            y_mask, y = self.mask(y)
        else:
            y, y_mask = self._y, self._y_mask
        if clone_kernel is True:
            kernel = self.kernel.clone_with_theta(theta)
        else:
            kernel = self.kernel
            kernel.theta = theta
        t = time.perf_counter()
        la = self._dense()
        on_device = self._device_gramian(la, kernel, X, eval_gradient,
                                         local_gradient)
        if on_device is not None:
            Kt, dKt = on_device
        elif eval_gradient is True:
            K, dK = self._gramian(self.alpha, X, kernel=kernel, jac=True)
            Kt, dKt = la.tensor(K), la.tensor(dK)
        else:
            Kt, dKt = la.tensor(self._gramian(self.alpha, X, kernel=kernel)), \
                None
        t_kernel = time.perf_counter() - t
        self._keep = None
        if not y_mask.all():
            keep = _torch().as_tensor(np.flatnonzero(y_mask),
                                      device=la.device)
            Kt = Kt.index_select(0, keep).index_select(1, keep)
            if dKt is not None and not hasattr(dKt, 'columns'):
                dKt = dKt.index_select(0, keep).index_select(1, keep)
            self._keep = keep          # (a LocalGradient is indexed in full)
        return theta, la, Kt, dKt, la.tensor(y), t_kernel

    def _device_gramian(self, la, kernel, X, jac, local_gradient=False):
        """Regularised float64 kernel matrix (and gradient over the active
        hyperparameters) as device tensors, straight from the kernel's device
        buffers -- if the kernel offers them (`device_gram` of the HIP
        marginalized graph kernel), the algebra runs on that GPU and no
        kernel options are in the way; None otherwise.  `local_gradient`:
        accept this rank's share of a pair-sharded gradient
        (`LocalGradient`) in place of the planes."""
        if la.device.type != 'cuda' or self.kernel_options \
                or not hasattr(kernel, 'device_gram'):
            return None
        torch = _torch()
        try:
            out = kernel.device_gram(X, eval_gradient=jac,
                                     **({'local_gradient': 'overlapped'}
                                        if jac and local_gradient else {}))
        except TypeError:            # not the HIP backend
            return None
        Kd, dKd = out if jac else (out, None)
        K = torch.as_tensor(Kd, device=la.device).to(torch.float64)
        diag = torch.diagonal(K)
        diag.copy_(self._regularize(diag, self.alpha))
        dK = None
        if hasattr(dKd, 'columns'):
            # this rank's pairs only: (pairs, n_dims) rows; the active
            # columns are picked where the rows are read (a pending gradient
            # must not be touched before the factorisation is enqueued)
            mask = np.asarray(kernel.active_theta_mask)
            if not mask.all():
                dKd.columns = np.flatnonzero(mask)
            dK = dKd
        elif dKd is not None:
            dK = torch.as_tensor(dKd, device=la.device)
            # a graph kernel hands over all its columns; transformers
            # (kernel/fix.py) already work on what the kernel protocol returns
            mask = np.asarray(kernel.active_theta_mask)
            if dK.shape[2] == len(mask) and not mask.all():
                dK = dK.index_select(2, torch.as_tensor(
                    np.flatnonzero(mask), device=la.device))
            dK = dK.to(torch.float64)
        return K, dK

    def log_marginal_likelihood(self, theta=None, X=None, y=None,
                                eval_gradient=False, clone_kernel=True,
                                verbose=False):
        """``y^T K^-1 y + log|K|`` at the log-scale hyperparameters `theta`
        (and its gradient w.r.t. `theta`)."""
        theta, la, K, dK, y, t_kernel = self._objective_inputs(
            theta, X, y, eval_gradient, clone_kernel, local_gradient=True)
        torch = _torch()
        t = time.perf_counter()
        grad = None
        done = False
        if la.native(K):
            # everything enqueued behind the one-launch factor-and-invert of
            # potrf.hip, then ONE download: the launch's status word and
            # log-determinant shares, y^T K^-1 y and the gradient's
            # contractions (round 5: three host synchronisations)
            from ._potrf import factor_inverse, parse_head, FactorisationError
            Kinv, head, nb = factor_inverse(K)
            Ky = Kinv @ y
            parts = [head[:16 + 2 * nb].view(torch.float64),
                     (y @ Ky).reshape(1)]
            if eval_gradient is True:
                W = Kinv - torch.outer(Ky, Ky)
                parts.append(_contract_local(W, dK, self._keep)
                             if hasattr(dK, 'columns')
                             else _contract_planes(W, dK))
            packed = torch.cat(parts).cpu().numpy()
            completed, logdet = parse_head(packed[:8 + nb], nb)
            if not completed:
                raise FactorisationError(
                    'spd_factor_invert_f64 gave up waiting for a tile')
            logdet *= 2.0
            if np.isfinite(logdet):       # (else: not positive definite)
                yKy = float(packed[8 + nb])
                if eval_gradient is True:
                    grad = packed[9 + nb:] * np.exp(theta)
                done = True
        if not done:
            Kinv, logdet = la.factor(K, self.beta, try_native=False)
            Ky = Kinv @ y
            yKy = float(y @ Ky)
            if eval_gradient is True:
                # tr(K^-1 dK_k) - (K^-1 y)^T dK_k (K^-1 y) = sum_ij W_ij dK_ijk
                # with the symmetric W = K^-1 - (K^-1 y)(K^-1 y)^T: one pass
                W = Kinv - torch.outer(Ky, Ky)
                if hasattr(dK, 'columns'):
                    # pair-sharded kernel: this rank's pairs + one all-reduce
                    d = _contract_local(W, dK, self._keep)
                else:
                    d = _contract_planes(W, dK)
                grad = d.cpu().numpy() * np.exp(theta)
        value = yKy + logdet
        t_linalg = time.perf_counter() - t
        if verbose:
            print(f'logP {value:12.5g}  y^T.K.y {yKy:12.5g}  '
                  f'log|K| {logdet:12.5g}  '
                  + (f'|dlogP| {np.linalg.norm(grad):12.5g}  '
                     if grad is not None else '')
                  + f't_kernel {t_kernel:8.2g} s  t_linalg {t_linalg:8.2g} s')
        self.last_timing = {'kernel': t_kernel, 'linalg': t_linalg}
        return (value, grad) if eval_gradient is True else value

    def squared_loocv_error(self, theta=None, X=None, y=None,
                            eval_gradient=False, clone_kernel=True,
                            verbose=False):
        """Half the sum of squared leave-one-out residuals
        ``e_i = (K^-1 y)_i / (K^-1)_ii`` (and its gradient w.r.t. `theta`)."""
        theta, la, K, dK, y, t_kernel = self._objective_inputs(
            theta, X, y, eval_gradient, clone_kernel)
        torch = _torch()
        t = time.perf_counter()
        Kinv, logdet = la.factor(K, self.beta)
        d = torch.diagonal(Kinv)
        Ky = Kinv @ y
        e = Ky / d
        value = 0.5 * float((e * e).sum())
        grad = None
        if eval_gradient is True:
            # d e_i = -[K^-1 dK K^-1 y]_i / d_i + e_i [K^-1 dK K^-1]_ii / d_i
            KdK = torch.einsum('ij,jlk->ilk', Kinv, dK)       # K^-1 dK_k
            first = torch.einsum('ilk,l->ik', KdK, Ky)        # K^-1 dK K^-1 y
            second = torch.einsum('ilk,li->ik', KdK, Kinv)    # diag(K^-1 dK K^-1)
            g = (-(e / d) @ first + (e * e / d) @ second)
            grad = g.cpu().numpy() * np.exp(theta)
        t_linalg = time.perf_counter() - t
        if verbose:
            print(f'Sq.Err. {value:12.5g}  log|K| {logdet:12.5g}  '
                  f't_kernel {t_kernel:8.2g} s  t_linalg {t_linalg:8.2g} s')
        self.last_timing = {'kernel': t_kernel, 'linalg': t_linalg}
        return (value, grad) if eval_gradient is True else value

    # -- persistence ---------------------------------------------------------------
    def save(self, path, filename='model.pkl', overwrite=False):
        """Pickle the trained state (without the kernel object; its
        hyperparameters are stored as `theta`)."""
        f_model = os.path.join(path, filename)
        if os.path.isfile(f_model) and not overwrite:
            raise RuntimeError(
                f'Path {f_model} already exists. To overwrite, set '
                '`overwrite=True`.')
        store = {k: v for k, v in self.__dict__.items()
                 if k not in ('kernel', '_la')}
        store['theta'] = np.array(self.kernel.theta)
        with open(f_model, 'wb') as f:
            pickle.dump(store, f, protocol=4)

    def load(self, path, filename='model.pkl'):
        with open(os.path.join(path, filename), 'rb') as f:
            store = pickle.load(f)
        theta = store.pop('theta')
        self.__dict__.update(**store)
        self.kernel.theta = theta
