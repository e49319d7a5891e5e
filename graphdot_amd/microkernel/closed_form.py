"""Microkernels that are written down directly (value, Jacobian and device
code as closed forms) rather than derived from a SymPy expression:
``Product``, ``DotProduct`` and ``KroneckerDelta``.

One declarative table (`_SPECS`) and one class factory replace a hand-written
class per kernel: a spec names the hyperparameters, gives the Python value and
Jacobian, the device expression templates and the value range.  Semantics as
in the reference (``graphdot/microkernel/product.py``, ``dotproduct.py``,
``kronecker_delta.py``); the expression strings are pinned by
``tests/golden/host_model.json``.
"""
import numpy as np
from ..codegen.cpptool import cpptype
from ..util.pretty_tuple import pretty_tuple
from ._base import MicroKernel


class _Spec:
    def __init__(self, name, params, value, jac, code, jac_code, span,
                 default_bounds=()):
        self.name, self.params = name, tuple(params)
        self.value, self.jac = value, jac            # Python closed forms
        self.code, self.jac_code = code, jac_code    # device templates
        self.span = span                             # theta -> (min, max)
        self.default_bounds = dict(default_bounds)


_SPECS = {
    'Product': _Spec(
        'Product', (),
        value=lambda a, b: a * b,
        jac=lambda a, b: np.array([]),
        code='({x} * {y})', jac_code=[],
        span=lambda: (None, None)),
    'DotProduct': _Spec(
        'DotProduct', (),
        value=lambda a, b: np.asarray(a) @ np.asarray(b),
        jac=lambda a, b: [],
        code='dotproduct({x}, {y})', jac_code=[],
        span=lambda: (0, np.inf)),
    'KroneckerDelta': _Spec(
        'KroneckerDelta', ('h',),
        value=lambda a, b, h: 1.0 if bool(a == b) else h,
        jac=lambda a, b, h: np.array([0.0 if bool(a == b) else 1.0]),
        code='({x} == {y} ? 1.0f : {h})',
        jac_code=['({x} == {y} ? 0.0f : 1.0f)'],
        span=lambda h: (h, 1),
        default_bounds={'h': (1e-3, 1)}),
}


def _build(spec):
    """The microkernel class of a spec (packed hyperparameters: float32)."""

    @cpptype([(p, np.float32) for p in spec.params])
    class ClosedForm(MicroKernel):
        _spec = spec

        def __init__(self, *values, **bounds):
            if len(values) != len(spec.params):
                raise TypeError(f'{spec.name} takes {len(spec.params)} '
                                'hyperparameter(s)')
            for p, v in zip(spec.params, values):
                setattr(self, p, float(v))
                b = bounds.pop(f'{p}_bounds', spec.default_bounds.get(p))
                setattr(self, f'{p}_bounds', b)
                self._assert_bounds(p, b)
            if bounds:
                raise TypeError(f'unknown arguments {sorted(bounds)}')

        @property
        def name(self):
            return spec.name

        def _values(self):
            return [getattr(self, p) for p in spec.params]

        def __call__(self, a, b, jac=False):
            f = spec.value(a, b, *self._values())
            return (f, spec.jac(a, b, *self._values())) if jac is True else f

        def __repr__(self):
            return f'{spec.name}({", ".join(map(str, self._values()))})'

        def gen_expr(self, x, y, theta_scope=''):
            names = {p: theta_scope + p for p in spec.params}
            return (spec.code.format(x=x, y=y, **names),
                    [j.format(x=x, y=y, **names) for j in spec.jac_code])

        @property
        def theta(self):
            if not spec.params:
                return tuple()
            return pretty_tuple(spec.name, spec.params)(*self._values())

        @theta.setter
        def theta(self, seq):
            for p, v in zip(spec.params, seq):
                setattr(self, p, v)

        @property
        def bounds(self):
            return tuple(getattr(self, f'{p}_bounds') for p in spec.params)

        @property
        def minmax(self):
            return spec.span(*self._values())

    ClosedForm.__name__ = ClosedForm.__qualname__ = spec.name + 'Kernel'
    return ClosedForm


#: ``Product()``: plain product of two scalar features (edge weights)
Product = _build(_SPECS['Product'])
_DotProduct = _build(_SPECS['DotProduct'])
_KroneckerDelta = _build(_SPECS['KroneckerDelta'])


def DotProduct():
    """Inner product of two vector-valued (variable-length) features; no
    hyperparameters."""
    return _DotProduct()


def KroneckerDelta(h, h_bounds=(1e-3, 1)):
    r""":math:`k_\delta(i, j) = 1` if :math:`i = j` else :math:`h`, with the
    similarity `h` of unequal features in (0, 1); `h_bounds` is
    ``(lower, upper)`` or ``'fixed'``."""
    return _KroneckerDelta(h, h_bounds=h_bounds)
