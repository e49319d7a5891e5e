"""Microkernels: positive-semidefinite functions between individual nodes and
edges, composable with ``+ * **`` and the feature-wise combinators below.
Same public names as ``graphdot.microkernel`` (reference
``graphdot/microkernel/__init__.py:7-37``)."""
from ._base import MicroKernel, Constant, Normalize
from .closed_form import Product, DotProduct, KroneckerDelta
from .square_exponential import SquareExponential
from .rational_quadratic import RationalQuadratic
from .composite import Composite, TensorProduct, Additive
from .convolution import Convolution

__all__ = [
    'MicroKernel', 'Product', 'Constant', 'KroneckerDelta',
    'SquareExponential', 'RationalQuadratic', 'Normalize', 'Composite',
    'TensorProduct', 'Additive', 'Convolution', 'DotProduct',
]
