"""Marginalized graph kernel: API surface of
``graphdot.kernel.marginalized`` on top of the MI355X HIP backend."""
from ._kernel import MarginalizedGraphKernel
from ._pairlist import AltMarginalizedGraphKernel
from ._backend import Backend
from ._backend_factory import backend_factory

__all__ = ['MarginalizedGraphKernel', 'AltMarginalizedGraphKernel',
           'Backend', 'backend_factory']
