import numpy as np
from ..codegen.cpptool import cpptype
from ..util.pretty_tuple import pretty_tuple
from ._base import MicroKernel


def Convolution(kernel: MicroKernel, mean=True):
    r"""Average (``mean=True``) or sum of a base microkernel over all pairs of
    elements of two variable-length feature sequences:
    :math:`k(X, Y) = \frac{1}{|X||Y|}\sum_{x\in X}\sum_{y\in Y}
    k_\mathrm{base}(x, y)` (reference
    ``graphdot/microkernel/convolution.py:10-96``)."""

    @cpptype(kernel=kernel.dtype)
    class ConvolutionOf(MicroKernel):
        @property
        def name(self):
            return 'Convolution'

        def __init__(self, kernel, mean):
            self.kernel = kernel
            self.mean = mean

        def __call__(self, X, Y, jac=False):
            reduce = np.mean if self.mean else np.sum
            if jac is True:
                F, J = zip(*[self.kernel(x, y, jac=True)
                             for x in X for y in Y])
                return reduce(F), reduce(np.asarray(J), axis=0)
            return reduce([self.kernel(x, y) for x in X for y in Y])

        def __repr__(self):
            return f'{self.name}({self.kernel!r})'

        def gen_expr(self, x, y, theta_scope=''):
            F, J = self.kernel.gen_expr('_1', '_2', theta_scope + 'kernel.')
            mean = 'true' if self.mean else 'false'
            f = (f'convolution<{mean}>([&](auto _1, auto _2)'
                 f'{{return {F};}}, {x}, {y})')
            jac = [(f'convolution_jacobian<{mean}>([&](auto _1, auto _2)'
                    f'{{return {j};}}, {x}, {y})') for j in J]
            return f, jac

        @property
        def theta(self):
            return pretty_tuple(self.name, ['base'])(self.kernel.theta)

        @theta.setter
        def theta(self, seq):
            self.kernel.theta = seq[0]

        @property
        def bounds(self):
            return (self.kernel.bounds,)

        @property
        def minmax(self):
            return self.kernel.minmax

    return ConvolutionOf(kernel, mean=mean)
