"""Microkernel object model: positive-definite functions on node/edge
features that know how to (a) evaluate themselves in Python, (b) print
themselves as a device C++ expression, and (c) describe their hyperparameters
as a packed struct.

Public behaviour follows the reference's ``graphdot/microkernel/_base.py:16-730``
(names, ``repr`` round-trips, ``theta``/``bounds``/``minmax``/``state``/``dtype``,
operator composition).  The printed expressions are consumed by the HIP JIT in
``kernel/marginalized/_backend_hip.py``.
"""
from abc import ABC, abstractmethod
from collections import OrderedDict
import numpy as np
import sympy as sy
from sympy.utilities.lambdify import lambdify
from ..codegen import Template
from ..codegen.cpptool import cpptype
from ..codegen.sympy_printer import hipcxxcode
from ..util.pretty_tuple import pretty_tuple


class MicroKernel(ABC):
    """Abstract base of all microkernels."""

    @property
    @abstractmethod
    def name(self):
        """Name of the kernel."""

    @abstractmethod
    def __call__(self, i, j, jac=False):
        """Value (and, if ``jac``, the gradient w.r.t. the hyperparameters)
        of the kernel on features `i`, `j`."""

    @abstractmethod
    def __repr__(self):
        """``eval(repr(k))`` must rebuild the kernel."""

    @abstractmethod
    def gen_expr(self, x, y, theta_scope=''):
        """Device expression for the kernel value and a list of expressions
        for its partial derivatives; `x`, `y` name the inputs and
        `theta_scope` prefixes hyperparameter names."""

    @property
    @abstractmethod
    def theta(self):
        """Tuple (tree) of hyperparameter values."""

    @theta.setter
    @abstractmethod
    def theta(self, value):
        pass

    @property
    @abstractmethod
    def bounds(self):
        """Tuple (tree) of (lower, upper) pairs or 'fixed'."""

    @property
    @abstractmethod
    def minmax(self):
        """(min, max) of the values the kernel can take."""

    @property
    def normalized(self):
        r""":math:`k(i,j)/\sqrt{k(i,i)k(j,j)}`"""
        return Normalize(self)

    def _assert_bounds(self, hyp, bounds):
        ok = bounds == 'fixed' if isinstance(bounds, str) else (
            isinstance(bounds, tuple) and len(bounds) == 2)
        if not ok:
            raise ValueError(
                f'Bounds for hyperparameter {hyp} of kernel {self.name} must '
                f'be a 2-tuple or "fixed": {bounds} provided.')

    @staticmethod
    def from_sympy(name, desc, expr, vars, *hyperparameter_specs,
                   minmax=(0, 1)):
        """Create a microkernel class from a SymPy expression of two input
        variables; all other symbols are hyperparameters, each specified as
        ``symbol`` | ``(symbol,)`` | ``(symbol, dtype)`` |
        ``(symbol, dtype, doc)`` | ``(symbol, dtype, lb, ub)`` |
        ``(symbol, dtype, lb, ub, doc)``."""
        return _from_sympy(name, desc, expr, vars, *hyperparameter_specs,
                           minmax=minmax)

    # -- algebra -------------------------------------------------------------
    def __add__(self, k):
        return MicroKernelExpr.add(self, k)

    def __radd__(self, k):
        return MicroKernelExpr.add(k, self)

    def __mul__(self, k):
        return MicroKernelExpr.mul(self, k)

    def __rmul__(self, k):
        return MicroKernelExpr.mul(k, self)

    def __pow__(self, c):
        return MicroKernelExpr.pow(self, c)


def _lift(k):
    return Constant(k) if np.isscalar(k) else k


class MicroKernelExpr(MicroKernel):
    """Binary expression node ``k1 <op> k2``."""

    opstr = None
    opname = None

    def __init__(self, k1, k2):
        self.k1 = k1
        self.k2 = k2

    @property
    def name(self):
        return self.opname

    def __repr__(self):
        return f'{self.k1!r} {self.opstr} {self.k2!r}'

    @property
    def theta(self):
        return pretty_tuple(self.name, ['lhs', 'rhs'])(
            self.k1.theta, self.k2.theta)

    @theta.setter
    def theta(self, seq):
        self.k1.theta, self.k2.theta = seq[0], seq[1]

    @property
    def bounds(self):
        return (self.k1.bounds, self.k2.bounds)

    def _sub_exprs(self, x, y, scope):
        return (self.k1.gen_expr(x, y, scope + 'k1.'),
                self.k2.gen_expr(x, y, scope + 'k2.'))

    @staticmethod
    def _make(base, k1, k2):
        cls = cpptype(k1=k1.dtype, k2=k2.dtype)(base)
        return cls(k1, k2)

    @staticmethod
    def add(k1, k2):
        return MicroKernelExpr._make(_Add, _lift(k1), _lift(k2))

    @staticmethod
    def mul(k1, k2):
        return MicroKernelExpr._make(_Multiply, _lift(k1), _lift(k2))

    @staticmethod
    def pow(k1, c):
        if np.isscalar(c):
            k2 = Constant(c)
        elif isinstance(c, MicroKernel) and c.name == 'Constant':
            k2 = c
        else:
            raise ValueError('Exponent must be a constant or constant '
                             f'microkernel, got {c} instead.')
        return MicroKernelExpr._make(_Exponentiation, k1, k2)


class _Add(MicroKernelExpr):
    opstr, opname = '+', 'Add'

    def __call__(self, i, j, jac=False):
        if jac is True:
            f1, J1 = self.k1(i, j, True)
            f2, J2 = self.k2(i, j, True)
            return f1 + f2, np.concatenate((np.atleast_1d(J1),
                                            np.atleast_1d(J2)))
        return self.k1(i, j) + self.k2(i, j)

    def gen_expr(self, x, y, theta_scope=''):
        (f1, J1), (f2, J2) = self._sub_exprs(x, y, theta_scope)
        return f'({f1} + {f2})', J1 + J2

    @property
    def minmax(self):
        (lo1, hi1), (lo2, hi2) = self.k1.minmax, self.k2.minmax
        return (lo1 + lo2, hi1 + hi2)


class _Multiply(MicroKernelExpr):
    opstr, opname = '*', 'Multiply'

    def __call__(self, i, j, jac=False):
        if jac is True:
            f1, J1 = self.k1(i, j, True)
            f2, J2 = self.k2(i, j, True)
            return f1 * f2, np.array([d * f2 for d in J1] +
                                     [f1 * d for d in J2])
        return self.k1(i, j) * self.k2(i, j)

    def gen_expr(self, x, y, theta_scope=''):
        (f1, J1), (f2, J2) = self._sub_exprs(x, y, theta_scope)
        return (f'({f1} * {f2})',
                [f'({d} * {f2})' for d in J1] + [f'({f1} * {d})' for d in J2])

    @property
    def minmax(self):
        (lo1, hi1), (lo2, hi2) = self.k1.minmax, self.k2.minmax
        return (lo1 * lo2, hi1 * hi2)


class _Exponentiation(MicroKernelExpr):
    opstr, opname = '**', 'Exponentiation'

    def __call__(self, i, j, jac=False):
        if jac is True:
            f1, J1 = self.k1(i, j, True)
            f2, J2 = self.k2(i, j, True)
            return f1**f2, np.array(
                [f2 * f1**(f2 - 1) * d for d in J1] +
                [f1**f2 * np.log(f1) * d for d in J2])
        return self.k1(i, j)**self.k2(i, j)

    def gen_expr(self, x, y, theta_scope=''):
        (f1, J1), (f2, J2) = self._sub_exprs(x, y, theta_scope)
        return (f'__powf({f1}, {f2})',
                [f'({f2} * __powf({f1}, {f2} - 1) * {d})' for d in J1] +
                [f'(__powf({f1}, {f2}) * __logf({f1}) * {d})' for d in J2])

    @property
    def minmax(self):
        (lo1, hi1), (lo2, hi2) = self.k1.minmax, self.k2.minmax
        return (lo1**lo2, hi1**hi2)


def Constant(c, c_bounds='fixed'):
    r"""A microkernel that ignores its inputs: :math:`k(\cdot,\cdot)\equiv c`,
    typically used as an adjustable weight in products.

    Parameters
    ----------
    c: float > 0
    c_bounds: (lower, upper) or 'fixed'
    """
    @cpptype(c=np.float32)
    class ConstantKernel(MicroKernel):
        @property
        def name(self):
            return 'Constant'

        def __init__(self, c, c_bounds):
            self.c = float(c)
            self.c_bounds = c_bounds
            self._assert_bounds('c', c_bounds)

        def __call__(self, i, j, jac=False):
            return (self.c, np.ones(1)) if jac is True else self.c

        def __repr__(self):
            return f'{self.name}({self.c})'

        def gen_expr(self, x, y, theta_scope=''):
            return f'{theta_scope}c', ['1.0f']

        @property
        def theta(self):
            return pretty_tuple(self.name, ['c'])(self.c)

        @theta.setter
        def theta(self, seq):
            self.c = seq[0]

        @property
        def bounds(self):
            return (self.c_bounds,)

        @property
        def minmax(self):
            return (self.c, self.c)

    return ConstantKernel(c, c_bounds)


def Normalize(kernel: MicroKernel):
    r"""Cosine normalisation :math:`k(x,y)/\sqrt{k(x,x)\,k(y,y)}`; the value
    is 0 whenever either self-similarity is not positive."""
    if kernel.name == 'Normalize':
        return kernel

    @cpptype(kernel=kernel.dtype)
    class Normalized(MicroKernel):
        @property
        def name(self):
            return 'Normalize'

        def __init__(self, kernel):
            self.kernel = kernel

        def __call__(self, X, Y, jac=False):
            k = self.kernel
            if jac is True:
                Fxx, Jxx = k(X, X, jac=True)
                Fxy, Jxy = k(X, Y, jac=True)
                Fyy, Jyy = k(Y, Y, jac=True)
                Jxx, Jxy, Jyy = map(np.asarray, (Jxx, Jxy, Jyy))
                if Fxx > 0 and Fyy > 0:
                    s = Fxx * Fyy
                    return (Fxy * s**-0.5,
                            Jxy * s**-0.5
                            - 0.5 * Fxy * s**-1.5 * (Jxx * Fyy + Fxx * Jyy))
                return 0.0, np.zeros_like(Jxy)
            Fxx, Fxy, Fyy = k(X, X), k(X, Y), k(Y, Y)
            if Fxx > 0 and Fyy > 0:
                return Fxy * (Fxx * Fyy)**-0.5
            return 0.0

        def __repr__(self):
            return f'{self.name}({self.kernel!r})'

        def gen_expr(self, x, y, theta_scope=''):
            F, J = self.kernel.gen_expr('_1', '_2', theta_scope + 'kernel.')
            f = (f'normalize([&](auto _1, auto _2){{return {F};}}, '
                 f'{x}, {y})')
            jac = [
                (f'normalize_jacobian([&](auto _1, auto _2){{return {F};}}, '
                 f'[&](auto _1, auto _2){{return {j};}}, {x}, {y})')
                for j in J]
            return f, jac

        @property
        def theta(self):
            return self.kernel.theta

        @theta.setter
        def theta(self, seq):
            self.kernel.theta = seq

        @property
        def bounds(self):
            return self.kernel.bounds

        @property
        def minmax(self):
            lo, hi = self.kernel.minmax
            return (lo / hi, 1)

    return Normalized(kernel)


def _parse_hyperspec(spec):
    if isinstance(spec, str) or not hasattr(spec, '__iter__'):
        spec = (spec,)
    spec = tuple(spec)
    f32 = np.dtype(np.float32)
    if len(spec) == 1:
        return spec[0], dict(dtype=f32)
    if len(spec) == 2:
        return spec[0], dict(dtype=np.dtype(spec[1]))
    if len(spec) == 3:
        return spec[0], dict(dtype=np.dtype(spec[1]), doc=spec[2])
    if len(spec) == 4:
        return spec[0], dict(dtype=np.dtype(spec[1]),
                             bounds=(spec[2], spec[3]))
    if len(spec) == 5:
        return spec[0], dict(dtype=np.dtype(spec[1]),
                             bounds=(spec[2], spec[3]), doc=spec[4])
    raise ValueError(
        'Invalid hyperparameter specification, must be one of\n'
        '(symbol)\n(symbol, dtype)\n(symbol, dtype, doc)\n'
        '(symbol, dtype, lb, ub)\n(symbol, dtype, lb, ub, doc)\n')


def _from_sympy(name, desc, expr, vars, *hyperparameter_specs, minmax=(0, 1)):
    assert isinstance(name, str) and name.isidentifier()
    if isinstance(expr, str):
        expr = sy.sympify(expr)
    if len(vars) != 2:
        raise ValueError('A microkernel must have exactly two variables')
    vars = [sy.Symbol(v) if isinstance(v, str) else v for v in vars]
    hyperdefs = OrderedDict(_parse_hyperspec(s) for s in hyperparameter_specs)
    kernel_name, kernel_minmax = name, minmax

    class _Meta(type(MicroKernel)):
        @property
        def dtype(cls):
            return cls._dtype

    class SymbolicKernel(MicroKernel, metaclass=_Meta):

        _expr = expr
        _vars = vars
        _hyperdefs = hyperdefs
        _dtype = np.dtype([(k, v['dtype']) for k, v in hyperdefs.items()],
                          align=True)

        @property
        def name(self):
            return kernel_name

        def __init__(self, *args, **kwargs):
            self._theta_values = values = OrderedDict()
            self._theta_bounds = bounds = OrderedDict()
            for symbol, value in zip(self._hyperdefs, args):
                values[symbol] = value
            for symbol, hdef in self._hyperdefs.items():
                if symbol in kwargs:
                    values[symbol] = kwargs[symbol]
                elif symbol not in values:
                    raise KeyError(f'Hyperparameter {symbol} not provided '
                                   f'for {self.name}')
                key = f'{symbol}_bounds'
                if key in kwargs:
                    bounds[symbol] = kwargs[key]
                elif 'bounds' in hdef:
                    bounds[symbol] = hdef['bounds']
                else:
                    raise KeyError(
                        f'Bounds for hyperparameter {symbol} of microkernel '
                        f'{self.name} not set, and no defaults were given.')
                self._assert_bounds(symbol, bounds[symbol])

        @classmethod
        def _compiled(cls):
            # lambdified value and partial derivatives, built once per class
            if '_lambdas' not in cls.__dict__:
                args = [*cls._vars, *cls._hyperdefs]
                cls._lambdas = (
                    lambdify(args, cls._expr),
                    [lambdify(args, sy.diff(cls._expr, h))
                     for h in cls._hyperdefs])
            return cls._lambdas

        def __call__(self, x1, x2, jac=False):
            fun, dfun = self._compiled()
            theta = tuple(self._theta_values.values())
            if jac is True:
                return (fun(x1, x2, *theta),
                        np.array([d(x1, x2, *theta) for d in dfun]))
            return fun(x1, x2, *theta)

        def __repr__(self):
            return Template('${cls}(${theta, }, ${bounds, })').render(
                cls=self.name,
                theta=[f'{n}={v}' for n, v in self._theta_values.items()],
                bounds=[f'{n}_bounds={v}'
                        for n, v in self._theta_bounds.items()])

        def gen_expr(self, x, y, theta_scope=''):
            # the strings depend on the class and the names only, never on
            # hyperparameter values: printed once (SymPy printing is ~1 ms
            # per expression, paid on every kernel evaluation otherwise)
            cache = type(self).__dict__.get('_expr_cache')
            if cache is None:
                cache = type(self)._expr_cache = {}
            key = (x, y, theta_scope)
            if key not in cache:
                nmap = {str(self._vars[0]): x, str(self._vars[1]): y}
                nmap.update({t: theta_scope + t for t in self._hyperdefs})
                cache[key] = (hipcxxcode(self._expr, nmap),
                              [hipcxxcode(sy.diff(self._expr, h), nmap)
                               for h in self._hyperdefs])
            fun, jac = cache[key]
            return fun, list(jac)

        @property
        def dtype(self):
            return self._dtype

        @property
        def state(self):
            return tuple(self._theta_values.values())

        @property
        def theta(self):
            return pretty_tuple(self.name, self._theta_values.keys())(
                **self._theta_values)

        @theta.setter
        def theta(self, seq):
            assert len(seq) == len(self._theta_values)
            for key, value in zip(self._hyperdefs, seq):
                self._theta_values[key] = value

        @property
        def bounds(self):
            return tuple(self._theta_bounds.values())

        @property
        def minmax(self):
            return kernel_minmax

    SymbolicKernel.__name__ = SymbolicKernel.__qualname__ = name
    doc = ['\n'.join(s.strip() for s in desc.split('\n')), '',
           'Parameters', '----------']
    for hname, hdef in hyperdefs.items():
        doc.append(f"{hname}: {hdef['dtype']}")
        doc += ['    ' + s.strip() for s in hdef.get('doc', '').split('\n')]
        doc.append(f'{hname}_bounds: tuple or "fixed"')
        doc.append(f'    Lower and upper bounds of `{hname}` for '
                   'hyperparameter optimization; "fixed" excludes it from '
                   'training.')
    SymbolicKernel.__doc__ = '\n'.join(doc)
    return SymbolicKernel
