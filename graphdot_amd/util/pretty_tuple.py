"""Named tuples that print one ``field : value`` per line, nested fields
indented -- the shape ``MarginalizedGraphKernel.hyperparameters`` returns
(reference: ``graphdot/util/pretty_tuple.py:7-30``)."""
from collections import namedtuple


def pretty_tuple(typename, fields):
    fields = list(fields)
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
