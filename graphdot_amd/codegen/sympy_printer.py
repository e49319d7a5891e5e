"""SymPy -> device C++ expression printer.

Emits the expression grammar of the reference's printer
(``graphdot/codegen/sympy_printer.py:7-44``): single-precision literals,
integer powers as ``graphdot::ipow<N>(x)`` / ``graphdot::ripow<N>(x)``,
everything else through the C++11 printer.  The device header
``device/fmath.h`` supplies those helpers for gfx950; a double-precision build
rewrites the float spellings afterwards (see ``to_real_expr``).
"""
import re
from sympy.codegen import ast
from sympy.printing.cxx import CXX11CodePrinter


class HIPCXXCodePrinter(CXX11CodePrinter):
    _ns = ''

    def __call__(self, expr, symbol_to_variable):
        self._symbol_map = symbol_to_variable
        return self.doprint(expr)

    def _print_Symbol(self, expr):
        name = self._symbol_map[super()._print_Symbol(expr)]
        if expr in self._settings['dereference']:
            return f'(*{name})'
        return name


def _is_int(e):
    return bool(e.is_integer)


hipcxxcode = HIPCXXCodePrinter(dict(
    user_functions={
        'Pow': [
            (lambda b, e: _is_int(e) and int(e) >= 0,
             lambda b, e: 'graphdot::ipow<%d>(%s)' % (int(e), b)),
            (lambda b, e: _is_int(e) and int(e) < 0,
             lambda b, e: 'graphdot::ripow<%d>(%s)' % (-int(e), b)),
            (lambda b, e: True, 'powf'),
        ]
    },
    type_aliases={ast.real: ast.float32, ast.integer: ast.int32},
))


_float_literal = re.compile(
    r'(?<![\w.])((?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?)[fF]\b')
_float_funcs = {
    'powf': 'pow', 'logf': 'log', 'expf': 'exp', 'sqrtf': 'sqrt',
    'rsqrtf': 'graphdot::rsqrt', 'fabsf': 'fabs', 'sinf': 'sin',
    'cosf': 'cos', 'tanhf': 'tanh', 'erff': 'erf', 'floorf': 'floor',
    'ceilf': 'ceil', 'fminf': 'fmin', 'fmaxf': 'fmax',
}
_float_func_re = re.compile(
    r'(?<![\w.])(' + '|'.join(sorted(map(re.escape, _float_funcs),
                                     key=len, reverse=True)) + r')\s*\(')


# the reference's fast-math intrinsics have no HIP spelling: both builds send
# them to the type-generic overloads of device/fmath.h
_intrinsics = {'__powf': 'graphdot::pow', '__logf': 'graphdot::log',
               '__expf': 'graphdot::exp'}
_intrinsic_re = re.compile(r'(?<![\w.])(__powf|__logf|__expf)\s*\(')
_expf_re = re.compile(r'(?<![\w.:])expf\s*\(')


def to_real_expr(expr, real='float32'):
    """Device spelling of a generated expression.  The grammar the
    microkernels print is the reference's (float32 literals, ``expf``,
    ``__powf``); the CUDA fast-math intrinsics among it are rewritten to the
    overloads of ``device/fmath.h`` for either arithmetic, and a float64
    build additionally drops the ``f`` literal suffixes and swaps
    ``expf``-style calls for their double overloads."""
    expr = _intrinsic_re.sub(lambda m: _intrinsics[m.group(1)] + '(', expr)
    if real == 'float32':
        # (expf through the overload of device/fmath.h, which spells it
        # exp2(x * log2 e) for the optimiser to fold the constant)
        return _expf_re.sub('graphdot::exp(', expr)
    expr = _float_literal.sub(
        lambda m: m.group(1) if any(c in m.group(1) for c in '.eE')
        else m.group(1) + '.0', expr)
    return _float_func_re.sub(lambda m: _float_funcs[m.group(1)] + '(', expr)
