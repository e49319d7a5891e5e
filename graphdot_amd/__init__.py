"""graphdot_amd -- MI355X-native marginalized graph kernel (GraphDot hot path).

Drop-in for the path ``graphdot.kernel.marginalized.MarginalizedGraphKernel``
-> backend -> device solver of yhtang/GraphDot, with hand-written HIP kernels
for gfx950 behind a C-ABI (``include/gdhip.h``).  See DESIGN.md.
"""
__version__ = '0.1.0'

from .graph import Graph

__all__ = ['Graph', '__version__']
