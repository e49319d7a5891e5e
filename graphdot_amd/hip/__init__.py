"""HIP runtime plumbing for the MI355X backend: ctypes binding of
``libgdhip.so`` (C ABI in ``include/gdhip.h``) and the hipcc JIT driver.
Takes the role of the reference's ``graphdot/cuda`` package (PyCUDA)."""
from .runtime import lib, HIPError, DeviceBuffer, device_props, ensure_device

__all__ = ['lib', 'HIPError', 'DeviceBuffer', 'device_props', 'ensure_device']
