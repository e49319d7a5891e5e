import numpy as np
from ._base import MicroKernel

SquareExponential = MicroKernel.from_sympy(
    'SquareExponential',
    r"""Square exponential (Gaussian/RBF) kernel
    :math:`k(x, y) = \exp(-\frac{1}{2}\frac{(x - y)^2}{\sigma^2})`,
    decaying smoothly from 1 to 0 with distance.""",
    'exp(-0.5 * (x - y)**2 * length_scale**-2)',
    ('x', 'y'),
    ('length_scale', np.float32, 1e-6, np.inf,
     r"""Decay length: the kernel is about 0.606 / 0.135 / 0.011 at one / two
     / three length scales."""),
    minmax=(0, 1)
)
