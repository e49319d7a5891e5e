"""Backend plugin seam (reference: ``graphdot/kernel/marginalized/_backend.py:6-9``).

A backend provides the static allocators ``array / zeros / empty`` that
``MarginalizedGraphKernel`` uses for its job list, offset table and output
buffers, and a ``__call__`` that fills those outputs::

    backend(graphs, node_kernel, edge_kernel, p, q, eps, ftol, gtol,
            jobs, starts, gramian, gradient, nX, nY, nJ, traits, timer)
"""
from abc import ABC, abstractmethod


class Backend(ABC):
    @abstractmethod
    def __call__(self):
        pass
