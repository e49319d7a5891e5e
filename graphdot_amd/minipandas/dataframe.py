"""Column-oriented table of :py:class:`Series`.

API subset of pandas that the graph container and the device-graph packer
need; behaviour follows ``graphdot/minipandas/dataframe.py:9-118`` -- in
particular ``rowtype()`` (packed, aligned struct dtype with the widest fields
first) which *defines* the node_t / edge_t layout on the device.
"""
import operator
from collections import namedtuple
import numpy as np
from .series import Series


_ROWTYPES = {}
_CONCRETE = operator.attrgetter('_concrete_type')


class DataFrame:

    def __init__(self, data=None):
        self._data = {}
        if isinstance(data, dict):
            for key, value in data.items():
                self[key] = value

    # -- column access -----------------------------------------------------
    def __getitem__(self, key):
        if isinstance(key, str):
            return self._data[key]
        if hasattr(key, '__iter__'):
            idx = np.asarray(key)
            if idx.dtype == np.bool_:
                return type(self)({k: v[idx] for k, v in self._data.items()})
            return type(self)({k: self._data[k] for k in key})
        raise TypeError(f'Invalid column index {key}')

    def __setitem__(self, key, value):
        self._data[key] = Series(value)

    def __getattr__(self, name):
        data = self.__dict__.get('_data')
        if data is not None and name in data:
            return data[name]
        raise AttributeError(f'Dataframe has no column {name}.')

    def __contains__(self, key):
        return key in self._data

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return max([0] + [len(col) for col in self._data.values()])

    def __repr__(self):
        return repr(self._data)

    @property
    def columns(self):
        return list(self._data)

    # -- row views ---------------------------------------------------------
    def rowtype(self, pack=True):
        """Aligned struct dtype of one row; with ``pack`` the fields are
        ordered by decreasing item size (stable) to minimise padding."""
        # (memoised on the columns' names and types: the graphs of a data set
        # share one row type, and building a struct dtype takes ~30 us)
        try:
            key = (pack, tuple(self._data),
                   tuple(map(_CONCRETE, self._data.values())))
            hit = _ROWTYPES.get(key)
        except TypeError:                  # an unhashable concrete type
            key = hit = None
        if hit is not None:
            return hit
        cols = self.columns
        ctypes = {k: np.dtype(self[k].concrete_type) for k in cols}
        if pack:
            order = np.argsort([-ctypes[k].itemsize for k in cols],
                               kind='stable')
            cols = [cols[i] for i in order]
        out = np.dtype([(k, ctypes[k].newbyteorder('=')) for k in cols],
                       align=True)
        if key is not None:
            if len(_ROWTYPES) > 256:
                _ROWTYPES.clear()
            _ROWTYPES[key] = out
        return out

    def rows(self, rowname='row'):
        """Iterate rows as named tuples (columns whose names are not valid
        identifiers, e.g. ``!i``, are skipped); rows also index by name."""
        visible = [k for k in self._data if k.isidentifier()]
        base = namedtuple(rowname, visible)

        def getitem(row, key):
            if isinstance(key, str):
                return getattr(row, key)
            return tuple.__getitem__(row, key)

        Row = type(rowname, (base,), {'__getitem__': getitem,
                                       '__slots__': ()})
        cols = [self._data[k] for k in visible]
        for i in range(len(self)):
            yield Row(*[c[i] for c in cols])

    def itertuples(self, tuplename='tuple'):
        yield from self.rows(rowname=tuplename)

    def iterrows(self, rowname='row'):
        yield from enumerate(self.rows(rowname=rowname))

    def iterstates(self, pack=True):
        """Rows as plain tuples in ``rowtype(pack)`` field order; non-scalar
        cells contribute their ``.state``."""
        names = list(self.rowtype(pack=pack).names)
        for row in zip(*[self[k] for k in names]):
            yield tuple(v if np.isscalar(v) else v.state for v in row)

    def to_pandas(self):
        import pandas as pd
        return pd.DataFrame({k: list(v) for k, v in self._data.items()})

    # -- structural ops ------------------------------------------------------
    def copy(self, deep=False):
        if deep:
            return type(self)({k: np.copy(v) for k, v in self._data.items()})
        return type(self)(self._data)

    def drop(self, keys, inplace=False):
        if inplace is True:
            for k in keys:
                del self._data[k]
            return None
        return self[[k for k in self.columns if k not in keys]]
