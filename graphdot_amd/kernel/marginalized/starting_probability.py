"""Starting probabilities of the random walk
(reference: ``graphdot/kernel/marginalized/starting_probability.py:9-140``).

Each class evaluates itself on a node table in Python (``__call__`` returns the
probabilities and their gradient) and prints itself as a device expression on
a node ``n`` (``gen_expr``)."""
from abc import ABC, abstractmethod
import numpy as np
from ...codegen.cpptool import cpptype
from ...util.pretty_tuple import pretty_tuple


class StartingProbability(ABC):
    """Assigns a non-negative (not necessarily normalised) starting
    probability to every node."""

    @abstractmethod
    def __call__(self, nodes):
        """nodes: DataFrame -> (p[n], dp[n_theta, n])"""

    @abstractmethod
    def gen_expr(self):
        """(device expression, [partial-derivative expressions])"""

    @property
    @abstractmethod
    def theta(self):
        pass

    @theta.setter
    @abstractmethod
    def theta(self, value):
        pass

    @property
    @abstractmethod
    def bounds(self):
        pass


@cpptype(p=np.float32)
class Uniform(StartingProbability):
    """The same starting probability `p` on every node.

    Parameters
    ----------
    p: float
    p_bounds: (lower, upper) or 'fixed'
    """

    def __init__(self, p, p_bounds=(1e-3, 1e3)):
        assert ((isinstance(p_bounds, tuple) and len(p_bounds) == 2)
                or p_bounds == 'fixed')
        self.p = p
        self.p_bounds = p_bounds

    def __call__(self, nodes):
        n = len(nodes)
        return self.p * np.ones(n), np.ones((1, n))

    def gen_expr(self):
        return 'p', ['1.f']

    @property
    def theta(self):
        return pretty_tuple('Uniform', ['p'])(self.p)

    @theta.setter
    def theta(self, t):
        self.p = t[0]

    @property
    def bounds(self):
        return (self.p_bounds,)


@cpptype(null=np.int8)
class Adhoc(StartingProbability):
    """A fixed (non-trainable) starting probability given twice: as a Python
    callable on the node table and as a C++ expression on a node ``n``."""
    null = 0

    def __init__(self, f, expr):
        self.f = f
        self.expr = expr

    def __call__(self, nodes):
        return self.f(nodes), np.empty((0, 0))

    def gen_expr(self):
        return f'({self.expr})', []

    @property
    def theta(self):
        return tuple()

    @theta.setter
    def theta(self, t):
        pass

    @property
    def bounds(self):
        return tuple()
