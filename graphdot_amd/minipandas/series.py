"""1-D column with a remembered element type.

Equivalent of the reference's ``graphdot/minipandas/series.py:7-38``:
scalar columns are stored with the smallest signed/float32+ dtype that holds
the values, anything else becomes an object column whose ``concrete_type`` is
the common Python class of its elements (or None when mixed).
"""
import numpy as np
from ..codegen.typetool import (
    common_concrete_type, common_min_type, is_scalar_type
)


class Series(np.ndarray):

    def __new__(cls, data):
        if isinstance(data, np.ndarray):
            s = data.view(cls)
            if is_scalar_type(s.dtype):
                s._concrete_type = np.dtype(s.dtype)
            else:
                s._concrete_type = common_concrete_type.of_values(data)
        else:
            data = list(data)
            t = common_min_type.of_values(data)
            dtype = np.dtype(t) if is_scalar_type(t) else np.dtype(object)
            s = np.empty(len(data), dtype=dtype).view(cls)
            for k, v in enumerate(data):   # element-wise keeps sequences 1-D
                s[k] = v
            s._concrete_type = np.dtype(t) if is_scalar_type(t) else t
        return s

    def __array_finalize__(self, obj):
        if obj is not None and not hasattr(self, '_concrete_type'):
            ct = getattr(obj, '_concrete_type', None)
            if ct is None and is_scalar_type(self.dtype):
                ct = np.dtype(self.dtype)
            self._concrete_type = ct

    def astype(self, dtype, *args, **kwargs):
        out = super().astype(dtype, *args, **kwargs)
        if is_scalar_type(out.dtype):
            out._concrete_type = np.dtype(out.dtype)
        return out

    def __repr__(self):
        return np.array2string(np.asarray(self), separator=',',
                               max_line_width=10**20)

    @property
    def concrete_type(self):
        return self._concrete_type

    def __reduce__(self):
        recon, args, state = super().__reduce__()
        return recon, args, (state, self.__dict__)

    def __setstate__(self, states):
        state, attrs = states
        self.__dict__.update(attrs)
        super().__setstate__(state)
