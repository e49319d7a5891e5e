"""Small host-side helpers shared by the marginalized-kernel path.

Mirrors the public names of the reference's ``graphdot/util/__init__.py:19-46``
(``Timer``) so that ``MarginalizedGraphKernel(..., timing=True)`` reports the
same per-phase table.
"""
import time
from collections import OrderedDict

__all__ = ['Timer']


class Timer:
    """Named stopwatch: ``tic(tag)`` / ``toc(tag)`` / ``report(unit)``."""

    _scale = {'s': 1.0, 'ms': 1e3, 'us': 1e6, 'ns': 1e9}

    def __init__(self):
        self.reset()

    def reset(self):
        self.t = OrderedDict()
        self.dt = OrderedDict()

    def tic(self, tag):
        self.t[tag] = time.perf_counter()

    def toc(self, tag):
        self.dt[tag] = time.perf_counter() - self.t.pop(tag)

    def report(self, unit='s'):
        try:
            scale = self._scale[unit]
        except KeyError:
            raise ValueError('Unknown unit %s' % unit)
        for tag, dt in self.dt.items():
            print('%9.1f %s on %s' % (dt * scale, unit, tag))
