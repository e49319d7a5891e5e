"""Python class <-> packed C++ struct bridge.

``@cpptype(field=dtype, ...)`` gives a class a numpy aligned-struct ``dtype``
(its device-side layout) and a ``state`` property (the tuple that fills one
struct instance); ``decltype(dtype)`` prints the matching C++ declaration.
Behaviour follows ``graphdot/codegen/cpptool.py:9-146``; the emitted type
names are the ones ``device/numpy_type.h`` defines.
"""
import itertools as it
import numpy as np
from .template import Template
from .typetool import can_cast, _dtype_util


def cpptype(decls=(), **kwdecls):
    ctype = np.dtype(list(decls) + list(kwdecls.items()), align=True)

    def decorate(cls):

        class CppType(type(cls)):
            @property
            def dtype(self):
                return ctype

            def __repr__(self):
                return f'@cpptype({ctype!r})\n{cls!r}'

        class Class(cls, metaclass=CppType):

            @property
            def dtype(self):
                return ctype

            @property
            def state(self):
                out = []
                for key in ctype.names:
                    field = ctype.fields[key][0]
                    value = getattr(self, key)
                    if _dtype_util.is_object(field):
                        out.append(value.state)
                    elif _dtype_util.is_array(field):
                        if not isinstance(value, np.ndarray):
                            raise TypeError(
                                f'attribute {key} declared as an array but '
                                f'the actual type is {type(value)}.')
                        if value.shape != field.shape:
                            raise ValueError(
                                f'attribute {key}: actual shape {value.shape} '
                                f'!= declared shape {field.shape}.')
                        if _dtype_util.is_object(field.base):
                            value = [value[c].state for c in
                                     it.product(*map(range, field.shape))]
                        out.append(np.array(value, dtype=field.base)
                                   .reshape(field.shape).tolist())
                    else:
                        out.append(field.type(value))
                return tuple(out)

            def __setattr__(self, name, value):
                if ctype.names and name in ctype.names:
                    lt = ctype.fields[name][0]
                    if _dtype_util.is_array(lt):
                        value = np.asarray(value)
                        if value.shape != lt.shape:
                            raise ValueError(
                                f"Cannot set array attribute '{name}' with "
                                f"value of mismatching shape:\n{value}")
                        if not can_cast(value.dtype, lt.base):
                            raise TypeError(
                                f"Cannot set attribute '{name}' "
                                f"(C++ type {decltype(lt)}) "
                                f"with values of {value.dtype}")
                    elif not _dtype_util.is_object(lt):
                        try:
                            rt = np.dtype(type(value))
                        except TypeError:
                            rt = None
                        if rt is None or not can_cast(rt, lt):
                            raise TypeError(
                                f"Cannot set attribute '{name}' "
                                f"(C++ type {decltype(lt)}) "
                                f"with value {value} of {type(value)}")
                super().__setattr__(name, value)

        Class.__name__ = cls.__name__
        Class.__qualname__ = getattr(cls, '__qualname__', cls.__name__)
        Class.__doc__ = cls.__doc__
        return Class

    return decorate


def _assert_is_identifier(name):
    if name and not name.isidentifier():
        raise ValueError(f'Name {name} is not a valid Python/C++ identifier.')


def decltype(t, name=''):
    """C++ declaration for numpy dtype `t` (optionally named `name`).

    A name of the form ``$key::template::<dtype.str>...`` declares a template
    instance, e.g. ``$rings::frozen_array::<i2`` -> ``frozen_array<int16>rings``
    (the mangling the device-graph packer uses for variable-length attributes).
    """
    t = np.dtype(t, align=True)
    if name.startswith('$'):
        n, tpl, *args = name[1:].split('::')
        _assert_is_identifier(n)
        return Template(r'${template}<${arguments,}>${name}').render(
            template=tpl, arguments=[decltype(a) for a in args], name=n)
    _assert_is_identifier(name)
    if _dtype_util.is_object(t):
        if len(t.names):
            return Template(r'struct{${members;};}${name}').render(
                name=name,
                members=[decltype(t.fields[v][0], v) for v in t.names])
        return f'constexpr static _empty {name} {{}}'
    if _dtype_util.is_array(t):
        return Template(r'${t} ${name}[${shape][}]').render(
            t=decltype(t.base), name=name, shape=t.shape)
    if t.kind == 'S':
        return f'char {name}[{t.itemsize}]'
    return f'{t.name} {name}'.strip()
