"""Gaussian process regression on the kernel protocol; mirrors
``graphdot.model.gaussian_process`` of the reference for the GPR class."""
try:      # torch's HIP runtime must be initialised before libgdhip's
    import torch as _torch   # (graphdot_amd.hip.runtime, _let_torch_initialise_first)
    _torch.cuda.is_available()
except ImportError:          # pragma: no cover
    pass
from .gpr import GaussianProcessRegressor

__all__ = ['GaussianProcessRegressor']
