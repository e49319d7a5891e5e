import numpy as np
from ..codegen.cpptool import cpptype
from ..util.pretty_tuple import pretty_tuple
from ._base import MicroKernel

_OPS = {
    '+': dict(ufunc=np.add, opname='Addtive',
              jfunc=lambda F, f, j: j,
              jgen=lambda F, j, i: j),
    '*': dict(ufunc=np.multiply, opname='Product',
              jfunc=lambda F, f, j: F / f * j,
              jgen=lambda F, j, i: '(' + ' * '.join(
                  F[:i] + (j,) + F[i + 1:]) + ')'),
}


def Composite(oper, **kw_kernels):
    r"""Microkernel over several named features, reducing per-feature
    microkernels with ``+`` or ``*``:
    :math:`k(X, Y) = k_{a_1}(X_{a_1}, Y_{a_1})\;\mathrm{op}\;
    k_{a_2}(X_{a_2}, Y_{a_2})\;\mathrm{op}\ldots`
    (reference ``graphdot/microkernel/composite.py:10-131``).

    Parameters
    ----------
    oper: '+' or '*'
    kw_kernels: attribute=microkernel pairs
    """
    if oper not in _OPS:
        raise ValueError(f'Invalid reduction operator {oper!r}.')
    # the class depends on the operator and the (attribute, state type) list
    # only: made once (building a class costs ~25 us, and the marginalized
    # kernel wraps its edge kernel in a TensorProduct on every evaluation of
    # weighted graphs)
    ckey = (oper, tuple((key, np.dtype(ker.dtype))
                        for key, ker in kw_kernels.items()))
    cls = _CLASSES.get(ckey)
    if cls is None:
        cls = _CLASSES[ckey] = _composite_class(oper, kw_kernels)
    return cls(oper, **kw_kernels)


_CLASSES = {}


def _composite_class(oper, kw_kernels):
    op = _OPS[oper]

    @cpptype([(key, ker.dtype) for key, ker in kw_kernels.items()])
    class CompositeKernel(MicroKernel):
        @property
        def name(self):
            return 'Composite'

        @property
        def opname(self):
            return op['opname']

        def __init__(self, opstr, **kw_kernels):
            self.opstr = opstr
            self.kw_kernels = kw_kernels

        def __repr__(self):
            args = ', '.join(f'{k}={K!r}' for k, K in self.kw_kernels.items())
            return f'{self.name}({self.opstr!r}, {args})'

        def __call__(self, X, Y, jac=False):
            if jac is True:
                F, J = zip(*[k(X[key], Y[key], True)
                             for key, k in self.kw_kernels.items()])
                S = op['ufunc'].reduce(F)
                return S, np.array([op['jfunc'](S, f, j)
                                    for f, js in zip(F, J) for j in js])
            return op['ufunc'].reduce(
                [k(X[key], Y[key]) for key, k in self.kw_kernels.items()])

        def gen_expr(self, x, y, theta_scope=''):
            F, J = zip(*[k.gen_expr(f'{x}.{key}', f'{y}.{key}',
                                    f'{theta_scope}{key}.')
                         for key, k in self.kw_kernels.items()])
            f = '(' + f' {self.opstr} '.join(F) + ')'
            return f, [op['jgen'](F, j, i)
                       for i, js in enumerate(J) for j in js]

        @property
        def theta(self):
            return pretty_tuple(self.name, self.kw_kernels.keys())(
                *[k.theta for k in self.kw_kernels.values()])

        @theta.setter
        def theta(self, seq):
            for k, value in zip(self.kw_kernels.values(), seq):
                k.theta = value

        @property
        def bounds(self):
            return pretty_tuple(self.name, self.kw_kernels.keys())(
                *[k.bounds for k in self.kw_kernels.values()])

        @property
        def minmax(self):
            return op['ufunc'].reduce(
                [k.minmax for k in self.kw_kernels.values()], axis=0)

    # sub-kernels as attributes: lets cpptype's `.state` walk the struct
    for key in kw_kernels:
        setattr(CompositeKernel, key,
                property(lambda self, key=key: self.kw_kernels[key]))

    return CompositeKernel


def TensorProduct(**kw_kernels):
    r""":math:`k_\otimes(X, Y) = \prod_a k_a(X_a, Y_a)`"""
    return Composite('*', **kw_kernels)


def Additive(**kw_kernels):
    r""":math:`k_\oplus(X, Y) = \sum_a k_a(X_a, Y_a)`"""
    return Composite('+', **kw_kernels)
