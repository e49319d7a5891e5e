import numpy as np
from ..codegen.cpptool import cpptype
from ._base import MicroKernel


def DotProduct():
    """Inner product of two vector-valued (variable-length) features; no
    hyperparameters (reference ``graphdot/microkernel/dotproduct.py``)."""

    @cpptype([])
    class DotProductKernel(MicroKernel):
        @property
        def name(self):
            return 'DotProduct'

        def __call__(self, X, Y, jac=False):
            v = np.asarray(X) @ np.asarray(Y)
            return (v, []) if jac is True else v

        def __repr__(self):
            return f'{self.name}()'

        def gen_expr(self, x, y, theta_scope=''):
            return f'dotproduct({x}, {y})', []

        @property
        def theta(self):
            return tuple()

        @theta.setter
        def theta(self, seq):
            pass

        @property
        def bounds(self):
            return tuple()

        @property
        def minmax(self):
            return (0, np.inf)

    return DotProductKernel()
