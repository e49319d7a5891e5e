"""Named tuples that print one ``field : value`` per line, nested fields
indented -- the shape ``MarginalizedGraphKernel.hyperparameters`` returns
(reference: ``graphdot/util/pretty_tuple.py:7-30``)."""
from collections import namedtuple
from functools import lru_cache


def pretty_tuple(typename, fields):
    # (building a namedtuple class costs ~70 us; `theta` properties ask for
    # the same few classes dozens of times per kernel evaluation)
    return _pretty_tuple(typename, tuple(fields))


@lru_cache(maxsize=None)
def _pretty_tuple(typename, fields):
    base = namedtuple(typename, fields)

    def _repr(self):
        lines = []
        for name in fields:
            value = getattr(self, name)
            if hasattr(value, '__iter__'):
                body = '\n\t'.join(repr(value).split('\n'))
                lines.append(f'{name} : {type(value).__name__}\n\t{body}')
            else:
                lines.append(f'{name} : {value!r}')
        return '\n'.join(lines)

    cls = type(typename, (base,), {'__repr__': _repr, '__slots__': ()})
    return cls
