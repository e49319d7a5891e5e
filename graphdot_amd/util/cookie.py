"""Per-graph cache that never survives pickling or deep copies.

Same contract as the reference's ``graphdot/util/cookie.py:5-12``: backends
park device images of a graph in ``graph.cookie[backend.uuid]``; a pickled or
deep-copied graph starts with an empty cookie so that stale device pointers
can never leak into another process or clone.
"""


class VolatileCookie(dict):

    def __reduce__(self):
        return (VolatileCookie, ())

    def __deepcopy__(self, memo):
        return VolatileCookie()
