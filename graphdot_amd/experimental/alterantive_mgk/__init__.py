"""Same import path (and spelling) as the reference's
``graphdot.experimental.alterantive_mgk``."""
from ...kernel.marginalized._pairlist import AltMarginalizedGraphKernel

__all__ = ['AltMarginalizedGraphKernel']
