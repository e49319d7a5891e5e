import numpy as np
from ._base import MicroKernel

RationalQuadratic = MicroKernel.from_sympy(
    'RationalQuadratic',
    r"""Rational quadratic kernel, a scale mixture of square exponential
    kernels; tends to the square exponential as alpha grows.""",
    '(1 + (x - y)**2 / (2 * alpha * length_scale**2))**(-alpha)',
    ('x', 'y'),
    ('length_scale', np.float32, 1e-6, np.inf,
     r"""Smallest length scale of the mixture."""),
    ('alpha', np.float32, 1e-3, np.inf,
     r"""Relative weight of large-scale components; larger values concentrate
     the mixture on short length scales."""),
    minmax=(0, 1)
)
