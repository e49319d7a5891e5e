"""Type inference for graph attributes.

Follows the promotion rules of the reference's
``graphdot/codegen/typetool.py:22-168``: integers go to the smallest *signed*
type that holds every value (so uint8 data become int16), floats are at least
float32, and non-scalar values report their Python class.
"""
import numpy as np

_convertible = {
    'b': 'b', 'i': 'iu', 'u': 'iu', 'f': 'f', 'c': 'c',
    'm': 'm', 'M': 'M', 'O': 'O', 'S': 'S', 'U': 'U', 'V': 'V',
}


def is_scalar_type(t):
    """True for numpy scalar dtypes/classes (bool, ints, floats, ...), False
    for Python classes such as list/tuple/ndarray and for object dtype."""
    try:
        dt = np.dtype(t)
    except TypeError:
        return False
    return dt.kind in 'biufcSUmM' and dt.names is None and dt.subdtype is None


def can_cast(src, dst):
    return np.dtype(src).kind in _convertible[np.dtype(dst).kind]


def _signed(t):
    if isinstance(t, np.dtype) and t.kind == 'u':
        return np.promote_types(t, np.int8)
    return t


def _merge(types, coerce, min_float, ensure_signed):
    t = None
    for r in types:
        if ensure_signed:
            r = _signed(r)
        if t is None:
            t = r
        elif t != r:
            if not coerce:
                return None
            t = np.promote_types(t, r)
    if isinstance(t, np.dtype) and t.kind == 'f':
        t = np.promote_types(t, min_float)
    return t


class common_min_type:

    @staticmethod
    def of_values(iterable, coerce=True, min_float=np.float32,
                  ensure_signed=True):
        """Smallest numpy type holding all values of `iterable`, or the common
        Python class for non-scalar values."""
        return _merge(
            (np.min_scalar_type(v) if np.isscalar(v) else type(v)
             for v in iterable),
            coerce, min_float, ensure_signed)

    @staticmethod
    def of_types(types, coerce=True, min_float=np.float32,
                 ensure_signed=True):
        return _merge(types, coerce, min_float, ensure_signed)


class common_concrete_type:

    @staticmethod
    def of_values(iterable):
        return common_concrete_type.of_types(map(type, iterable))

    @staticmethod
    def of_types(types):
        t = None
        for i in types:
            if t is None:
                t = i
            elif t != i:
                return None
        return t


def have_same_fields(t1, t2):
    if bool(t1.fields) != bool(t2.fields):
        return False
    if t1.fields:
        if set(t1.fields) != set(t2.fields):
            return False
        return all(have_same_fields(t1.fields[f][0], t2.fields[f][0])
                   for f in t1.fields)
    return True


class _dtype_util:
    @staticmethod
    def is_object(t):
        return t.names is not None

    @staticmethod
    def is_array(t):
        return t.subdtype is not None
