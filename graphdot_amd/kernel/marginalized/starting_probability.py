"""Starting probabilities of the random walk.

A starting probability gives every node a non-negative (not necessarily
normalised) weight.  It exists twice: as a Python function of the node table
(used by oracles and by the host-side part of the nodal gradient) and as a
device expression on one node ``n`` that the code generator pastes into the
solver (``HIPBackend.gencode_probability``).  Interface and semantics follow
the reference (``graphdot/kernel/marginalized/starting_probability.py:9-140``):
``p(nodes) -> (values[n], gradient[n_theta, n])``, ``p.gen_expr() -> (expr,
[d expr / d theta])``, log-scale-ready ``theta`` / ``bounds``.
"""
import numpy as np
from ...codegen.cpptool import cpptype
from ...util.pretty_tuple import pretty_tuple


class StartingProbability:
    """Base: a parameter-free starting probability.  Subclasses list their
    trainable parameters in ``_trainable`` (attribute names; ``<name>_bounds``
    holds the bounds) and implement ``__call__`` and ``gen_expr``."""

    _trainable = ()

    def __call__(self, nodes):
        raise NotImplementedError

    def gen_expr(self):
        raise NotImplementedError

    @property
    def theta(self):
        if not self._trainable:
            return tuple()
        return pretty_tuple(type(self).__name__, self._trainable)(
            *[getattr(self, k) for k in self._trainable])

    @theta.setter
    def theta(self, values):
        for k, v in zip(self._trainable, values):
            setattr(self, k, v)

    @property
    def bounds(self):
        return tuple(getattr(self, f'{k}_bounds') for k in self._trainable)


@cpptype(p=np.float32)
class Uniform(StartingProbability):
    """One starting probability `p` for all nodes; `p_bounds` is
    ``(lower, upper)`` or ``'fixed'``."""

    _trainable = ('p',)

    def __init__(self, p, p_bounds=(1e-3, 1e3)):
        if p_bounds != 'fixed' and not (
                isinstance(p_bounds, tuple) and len(p_bounds) == 2):
            raise AssertionError(f'invalid bounds {p_bounds!r}')
        self.p, self.p_bounds = p, p_bounds

    def __call__(self, nodes):
        one = np.ones(len(nodes))
        return self.p * one, one[None, :]

    def gen_expr(self):
        return 'p', ['1.f']


@cpptype(null=np.int8)
class Adhoc(StartingProbability):
    """A fixed starting probability given twice by the user: `f`, a Python
    callable on the node table, and `expr`, the same thing as a C++
    expression on a node ``n``.  Nothing to train."""

    null = 0          # the packed state needs one byte

    def __init__(self, f, expr):
        self.f, self.expr = f, expr

    def __call__(self, nodes):
        return self.f(nodes), np.empty((0, 0))

    def gen_expr(self):
        return f'({self.expr})', []
