"""Per-graph cache that never survives pickling or deep copies.

Same contract as the reference's ``graphdot/util/cookie.py:5-12``: backends
park device images of a graph in ``graph.cookie[backend.uuid]``; a pickled or
deep-copied graph starts with an empty cookie so that stale device pointers
can never leak into another process or clone.
"""


from collections import OrderedDict


class VolatileCookie(dict):
    #: bumped whenever an entry leaves any cookie (a graph was permuted,
    #: re-typed, or a packing was dropped): what `IdentityCache` entries are
    #: valid against
    epoch = 0

    def __reduce__(self):
        return (VolatileCookie, ())

    def __deepcopy__(self, memo):
        return VolatileCookie()

    def clear(self):
        VolatileCookie.epoch += 1
        super().clear()

    def __delitem__(self, key):
        VolatileCookie.epoch += 1
        super().__delitem__(key)

    def pop(self, *args):
        VolatileCookie.epoch += 1
        return super().pop(*args)

    def popitem(self):
        VolatileCookie.epoch += 1
        return super().popitem()


class IdentityCache:
    """What was derived from a *list of graphs* the last few times, keyed by
    the identities of its members: a training loop passes the same thousand
    graphs every call, and per-graph Python work (cookie lookups, row-type
    comparisons) is then most of the host time of a call.  An entry holds its
    graphs (so that the identities stay theirs) and dies with the cookie
    epoch, i.e. as soon as any graph drops cached state."""

    def __init__(self, maxsize=4):
        self.maxsize = maxsize
        self._entries = OrderedDict()

    def get(self, graphs):
        key = tuple(map(id, graphs))
        hit = self._entries.get(key)
        if hit is not None and hit[0] == VolatileCookie.epoch:
            self._entries.move_to_end(key)
            return key, hit[1]
        return key, None

    def put(self, key, graphs, value):
        self._entries[key] = (VolatileCookie.epoch, value, tuple(graphs))
        while len(self._entries) > self.maxsize:
            self._entries.popitem(last=False)
