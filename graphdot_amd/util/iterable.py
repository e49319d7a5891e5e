"""Tree <-> flat-sequence helpers used by the hyperparameter plumbing.

API-compatible with the reference's ``graphdot/util/iterable.py:5-37``
(``flatten``, ``fold_like``, ``replace``).
"""


def flatten(tree):
    """Depth-first leaves of a nest of lists/tuples: ((1, 2), 3) -> 1, 2, 3."""
    stack = [iter(tree)]
    while stack:
        for item in stack[-1]:
            if isinstance(item, (list, tuple)):
                stack.append(iter(item))
                break
            yield item
        else:
            stack.pop()


def _n_leaves(node):
    return sum(1 for _ in flatten(node))


def fold_like(flat, example):
    """Re-nest the flat sequence so that it has the shape of `example`:
    fold_like([1, 2, 3], ((None, None), None)) -> ((1, 2), 3)."""
    out, pos = [], 0
    for item in example:
        if hasattr(item, '__iter__'):
            n = _n_leaves(item)
            out.append(fold_like(flat[pos:pos + n], item))
            pos += n
        else:
            out.append(flat[pos])
            pos += 1
    return tuple(out)


def replace(iterable, old, new):
    """Yield the items of `iterable` with every `old` swapped for `new`."""
    for item in iterable:
        # `==` on ndarray/tuple mixes is avoided on purpose: only scalars and
        # strings (e.g. 'fixed') are ever searched for.
        if isinstance(item, type(old)) and item == old:
            yield new
        else:
            yield item
