"""Backend selection (reference:
``graphdot/kernel/marginalized/_backend_factory.py:6-18``).

``'auto'`` and ``'hip'`` build a :py:class:`HIPBackend`; ``'cuda'`` is accepted
as an alias so that reference scripts run unchanged.  There is deliberately no
CPU fallback: if the HIP library or device is missing this raises.
"""
from ._backend import Backend


def backend_factory(backend, *args, **kwargs):
    if isinstance(backend, Backend):
        return backend
    if backend in ('hip', 'cuda'):
        from ._backend_hip import HIPBackend
        return HIPBackend(*args, **kwargs)
    if backend == 'auto':
        try:
            from ._backend_hip import HIPBackend
            return HIPBackend(*args, **kwargs)
        except Exception as e:
            raise RuntimeError(f'Cannot auto-select backend: {e}')
    raise ValueError(f'Unknown backend {backend}')
