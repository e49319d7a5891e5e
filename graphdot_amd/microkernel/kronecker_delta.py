import numpy as np
from ..codegen.cpptool import cpptype
from ..util.pretty_tuple import pretty_tuple
from ._base import MicroKernel


def KroneckerDelta(h, h_bounds=(1e-3, 1)):
    r""":math:`k_\delta(i, j) = 1` if :math:`i = j` else :math:`h`.

    Parameters
    ----------
    h: float in (0, 1)
        Similarity assigned to unequal features.
    h_bounds: (lower, upper) or 'fixed'
    (reference: ``graphdot/microkernel/kronecker_delta.py:9-72``)
    """
    @cpptype(h=np.float32)
    class KroneckerDeltaKernel(MicroKernel):
        @property
        def name(self):
            return 'KroneckerDelta'

        def __init__(self, h, h_bounds):
            self.h = float(h)
            self.h_bounds = h_bounds
            self._assert_bounds('h', h_bounds)

        def __call__(self, i, j, jac=False):
            same = bool(i == j)
            f = 1.0 if same else self.h
            if jac is True:
                return f, np.array([0.0 if same else 1.0])
            return f

        def __repr__(self):
            return f'{self.name}({self.h})'

        def gen_expr(self, x, y, theta_scope=''):
            return (f'({x} == {y} ? 1.0f : {theta_scope}h)',
                    [f'({x} == {y} ? 0.0f : 1.0f)'])

        @property
        def theta(self):
            return pretty_tuple(self.name, ['h'])(self.h)

        @theta.setter
        def theta(self, seq):
            self.h = seq[0]

        @property
        def bounds(self):
            return (self.h_bounds,)

        @property
        def minmax(self):
            return (self.h, 1)

    return KroneckerDeltaKernel(h, h_bounds)
