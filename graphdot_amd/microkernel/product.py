import numpy as np
from ..codegen.cpptool import cpptype
from ._base import MicroKernel


@cpptype([])
class Product(MicroKernel):
    """Plain product of two scalar features, ``x * y`` (used for edge
    weights; reference ``graphdot/microkernel/product.py``)."""

    @property
    def name(self):
        return 'Product'

    def __call__(self, x1, x2, jac=False):
        return (x1 * x2, np.array([])) if jac is True else x1 * x2

    def __repr__(self):
        return f'{self.name}()'

    def gen_expr(self, x, y, theta_scope=''):
        return f'({x} * {y})', []

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
        return (None, None)
