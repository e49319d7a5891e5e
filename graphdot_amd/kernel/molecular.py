"""Ready-made marginalized graph kernel for 3-D molecular structures
(Tang & de Jong, J. Chem. Phys. 150, 044107 (2019)); the class of the
reference's ``graphdot/kernel/molecular.py:11-91``: elements compared with a
Kronecker delta, inter-atomic distances with a Gaussian, on graphs that carry
an ``element`` node attribute and a ``length`` edge attribute."""
import copy
from .marginalized import MarginalizedGraphKernel
from ..microkernel import KroneckerDelta, SquareExponential, TensorProduct


class Tang2019MolecularKernel:
    """
    Parameters
    ----------
    stopping_probability: float in (0, 1)
    starting_probability: float, or anything ``p=`` of the graph kernel takes
    element_prior: float in (0, 1)
        Similarity of two different elements (equal elements: 1).
    edge_length_scale: float > 0
        Length scale of the Gaussian on edge lengths.
    kwargs: passed to MarginalizedGraphKernel (e.g. ``backend=``)
    """

    def __init__(self, stopping_probability=0.01, starting_probability=1.0,
                 element_prior=0.2, edge_length_scale=0.05, **kwargs):
        self.stopping_probability = stopping_probability
        self.starting_probability = starting_probability
        self.element_prior = element_prior
        self.edge_length_scale = edge_length_scale
        self.kernel = MarginalizedGraphKernel(
            TensorProduct(element=KroneckerDelta(element_prior)),
            TensorProduct(length=SquareExponential(edge_length_scale)),
            q=stopping_probability, p=starting_probability, **kwargs)

    def __call__(self, X, Y=None, **kwargs):
        return self.kernel(X, Y, **kwargs)

    def diag(self, X, **kwargs):
        return self.kernel.diag(X, **kwargs)

    def __getattr__(self, name):
        # hyperparameters, theta, bounds, hyperparameter_bounds, ...
        if name == 'kernel':
            raise AttributeError(name)
        return getattr(self.kernel, name)

    @property
    def theta(self):
        return self.kernel.theta

    @theta.setter
    def theta(self, value):
        self.kernel.theta = value

    def clone_with_theta(self, theta):
        clone = copy.deepcopy(self)
        clone.theta = theta
        return clone
