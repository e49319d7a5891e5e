"""Kernel transformers on top of the kernel protocol: cosine normalisation
and exponentiation (behaviour of the reference's ``graphdot/kernel/fix.py:
7-215``).  They only post-process ``kernel(X, Y, eval_gradient)`` and
``kernel.diag(X)`` with numpy, so they work with any kernel that follows the
protocol, the HIP marginalized graph kernel included.

A shared base forwards the hyperparameter interface; a transformer implements
`_apply` (values) and `_apply_jac` (values + gradient) on the raw matrix and
the two diagonals.
"""
import copy
import numpy as np
from ..util.pretty_tuple import pretty_tuple


class _Transformed:
    """kernel -> kernel, same protocol."""

    def __init__(self, kernel):
        self.kernel = kernel

    # -- hyperparameter interface: the wrapped kernel's by default -----------
    @property
    def hyperparameters(self):
        return self.kernel.hyperparameters

    @property
    def theta(self):
        return self.kernel.theta

    @theta.setter
    def theta(self, value):
        self.kernel.theta = value

    @property
    def hyperparameter_bounds(self):
        return self.kernel.hyperparameter_bounds

    @property
    def bounds(self):
        return self.kernel.bounds

    def clone_with_theta(self, theta):
        clone = copy.deepcopy(self)
        clone.theta = theta
        return clone

    # -- optional device path (see MarginalizedGraphKernel.device_gram) --------
    @property
    def active_theta_mask(self):
        return self.kernel.active_theta_mask

    def _inner_device_gram(self, X, eval_gradient):
        """float64 torch tensors (K, dK over all hyperparameter columns) of
        the wrapped kernel on the GPU; TypeError if it has no device path."""
        inner = getattr(self.kernel, 'device_gram', None)
        if inner is None:
            raise TypeError('the wrapped kernel has no device_gram')
        import torch
        out = inner(X, eval_gradient=eval_gradient)
        R, dR = out if eval_gradient else (out, None)
        R = torch.as_tensor(R, device='cuda').to(torch.float64)
        if dR is not None:
            dR = torch.as_tensor(dR, device='cuda').to(torch.float64)
        return R, dR


class Normalization(_Transformed):
    r""":math:`k_n(x, y) = k(x, y) / \sqrt{k(x, x)\,k(y, y)}`."""

    def __call__(self, X, Y=None, eval_gradient=False, **options):
        k = self.kernel
        if eval_gradient is True:
            R, dR = k(X, Y, eval_gradient=True, **options)
            if Y is None:
                dl, ddl = R.diagonal(), np.einsum('iik->ik', dR)
                dr, ddr = dl, ddl
            else:
                dl, ddl = k.diag(X, True, **options)
                dr, ddr = k.diag(Y, True, **options)
        else:
            R = k(X, Y, **options)
            if Y is None:
                dl = dr = R.diagonal()
            else:
                dl, dr = k.diag(X, **options), k.diag(Y, **options)
        sl, sr = dl**-0.5, dr**-0.5
        K = sl[:, None] * R * sr[None, :]
        if eval_gradient is not True:
            return K
        # d(R / sqrt(a b)) = dR / sqrt(a b) - K (da / a + db / b) / 2
        dK = (sl[:, None, None] * dR * sr[None, :, None]
              - 0.5 * K[:, :, None] * ((ddl / dl[:, None])[:, None, :]
                                       + (ddr / dr[:, None])[None, :, :]))
        return K, np.asfortranarray(dK)

    def device_gram(self, X, eval_gradient=False, local_gradient=False):
        """`__call__(X)` computed on the GPU from the wrapped kernel's device
        buffers; returns torch tensors.  (`local_gradient`: accepted for the
        regressor's call and not forwarded -- the transformation needs the
        whole gradient planes, not one rank's pairs.)"""
        R, dR = self._inner_device_gram(X, eval_gradient)
        d = R.diagonal()
        s = d.rsqrt()
        K = s[:, None] * R * s[None, :]
        if not eval_gradient:
            return K
        rel = dR.diagonal(dim1=0, dim2=1).T / d[:, None]      # d log k(x, x)
        dK = (s[:, None, None] * dR * s[None, :, None]
              - 0.5 * K[:, :, None] * (rel[:, None, :] + rel[None, :, :]))
        return K, dK

    def diag(self, X, eval_gradient=False, **options):
        """Ones (and, like the reference, ones for the 'gradient')."""
        one = np.ones(len(X))
        if eval_gradient is True:
            return one, np.ones((len(X), len(self.kernel.theta)))
        return one


class Exponentiation(_Transformed):
    r""":math:`k_\xi(x, y) = k(x, y)^\xi`; the exponent is the first
    hyperparameter (`xi_bounds`: its search range)."""

    def __init__(self, kernel, xi=1.0, xi_bounds=(0.1, 20.0)):
        super().__init__(kernel)
        self.xi = xi
        self.xi_bounds = xi_bounds

    def __call__(self, X, Y=None, eval_gradient=False, **options):
        if eval_gradient is not True:
            return self.kernel(X, Y, **options)**self.xi
        R, dR = self.kernel(X, Y, eval_gradient=True, **options)
        K = R**self.xi
        # columns: d/d xi = K log R, then xi R^(xi - 1) dR/d theta
        dK = np.concatenate(((K * np.log(R))[:, :, None],
                             (self.xi * R**(self.xi - 1))[:, :, None] * dR),
                            axis=2)
        return K, dK

    def device_gram(self, X, eval_gradient=False, local_gradient=False):
        """(`local_gradient` is accepted and not forwarded: see
        `Normalization.device_gram`.)"""
        import torch
        R, dR = self._inner_device_gram(X, eval_gradient)
        K = R**self.xi
        if not eval_gradient:
            return K
        return K, torch.cat(((K * R.log())[:, :, None],
                             (self.xi * R**(self.xi - 1))[:, :, None] * dR),
                            dim=2)

    @property
    def active_theta_mask(self):
        return np.concatenate(([True], self.kernel.active_theta_mask))

    def diag(self, X, eval_gradient=False, **options):
        """``kernel.diag(X) ** xi`` (with the gradient, which the reference's
        class does not offer, so that Normalization can wrap this one for
        X-versus-Y evaluations too)."""
        if eval_gradient is not True:
            return self.kernel.diag(X, **options)**self.xi
        d, dd = self.kernel.diag(X, True, **options)
        k = d**self.xi
        return k, np.concatenate(((k * np.log(d))[:, None],
                                  (self.xi * d**(self.xi - 1))[:, None] * dd),
                                 axis=1)

    @property
    def hyperparameters(self):
        return pretty_tuple('Exponentiation', ['xi', 'kernel'])(
            self.xi, self.kernel.hyperparameters)

    @property
    def theta(self):
        return np.concatenate((np.log([self.xi]), self.kernel.theta))

    @theta.setter
    def theta(self, value):
        self.xi = float(np.exp(value[0]))
        self.kernel.theta = value[1:]

    @property
    def hyperparameter_bounds(self):
        return pretty_tuple('Exponentiation', ['xi', 'kernel'])(
            self.xi_bounds, self.kernel.hyperparameter_bounds)

    @property
    def bounds(self):
        return np.vstack((np.log([self.xi_bounds]), self.kernel.bounds))
