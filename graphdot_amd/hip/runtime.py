"""ctypes binding of libgdhip.so (declared in include/gdhip.h).

There is no CPU fallback: if the shared library is missing it is built with
hipcc, and if no gfx950 device is present every device call raises HIPError.
"""
import ctypes
import os
import subprocess
import threading

_here = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_here), 'csrc')
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(_here)), 'include')
LIB_PATH = os.path.join(CSRC, 'libgdhip.so')

_lock = threading.Lock()
_thread = threading.local()      # .device: the device current in this thread
_lib = None
_device = None


class HIPError(RuntimeError):
    pass


class DeviceProps(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 256), ('arch', ctypes.c_char * 64),
                ('compute_units', ctypes.c_int32),
                ('wavefront_size', ctypes.c_int32),
                ('max_threads_per_block', ctypes.c_int32),
                ('clock_khz', ctypes.c_int32),
                ('lds_per_block', ctypes.c_int64),
                ('total_mem', ctypes.c_int64)]


# every exported symbol of include/gdhip.h with its argument types
_vp, _sz, _u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
_P = ctypes.POINTER
SIGNATURES = {
    'gd_device_count': [_P(ctypes.c_int)],
    'gd_init': [ctypes.c_int],
    'gd_set_device': [ctypes.c_int],
    'gd_device_props': [ctypes.c_int, _P(DeviceProps)],
    'gd_device_sync': [],
    'gd_malloc': [_P(_vp), _sz],
    'gd_free': [_vp],
    'gd_host_alloc': [_P(_vp), _sz],
    'gd_host_free': [_vp],
    'gd_memcpy_h2d': [_vp, _vp, _sz, _vp],
    'gd_memcpy_d2h': [_vp, _vp, _sz, _vp],
    'gd_memcpy_d2d': [_vp, _vp, _sz, _vp],
    'gd_memset': [_vp, ctypes.c_int, _sz, _vp],
    'gd_module_load': [_vp, _sz, _P(_vp)],
    'gd_module_unload': [_vp],
    'gd_module_get_function': [_vp, ctypes.c_char_p, _P(_vp)],
    'gd_module_get_global': [_vp, ctypes.c_char_p, _P(_vp), _P(_sz)],
    'gd_function_attributes': [_vp, _P(ctypes.c_int), _P(ctypes.c_int),
                               _P(ctypes.c_int)],
    'gd_function_set_max_dynamic_lds': [_vp, ctypes.c_int],
    'gd_launch': [_vp, _u32, _u32, _u32, _vp, _vp, _sz],
    'gd_launch_cooperative': [_vp, _u32, _u32, _u32, _vp, _vp, _sz],
    'gd_function_max_active_blocks': [_vp, _u32, _u32, _P(ctypes.c_int)],
    'gd_stream_create': [_P(_vp)],
    'gd_stream_create_low_priority': [_P(_vp)],
    'gd_stream_destroy': [_vp],
    'gd_stream_sync': [_vp],
    'gd_event_create': [_P(_vp)],
    'gd_event_destroy': [_vp],
    'gd_event_record': [_vp, _vp],
    'gd_event_sync': [_vp],
    'gd_stream_wait_event': [_vp, _vp],
    'gd_event_elapsed_ms': [_vp, _vp, _P(ctypes.c_float)],
    'gd_comm_unique_id': [_vp],
    'gd_comm_init_rank': [_P(_vp), ctypes.c_int, _vp, ctypes.c_int],
    'gd_comm_destroy': [_vp],
    'gd_all_gather': [_vp, _vp, _sz, ctypes.c_int, _vp, _vp],
}


def build_library(force=False):
    """hipcc-build csrc/gdhip.cpp -> csrc/libgdhip.so (host code only)."""
    src = os.path.join(CSRC, 'gdhip.cpp')
    hdr = os.path.join(INCLUDE, 'gdhip.h')
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src),
                                                  os.path.getmtime(hdr))):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '-O2', '-fPIC', '-shared', '-std=c++17', f'-I{INCLUDE}',
           src, '-o', LIB_PATH, '-ldl']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise HIPError(f'building libgdhip.so failed:\n{r.stderr}')
    return LIB_PATH


def _let_torch_initialise_first():
    """PyTorch-ROCm ships its own copy of the HIP runtime.  If that copy is
    initialised after this library's, torch reports no usable GPU
    (`torch.cuda.is_available()` is False; measured on this image), the
    other order works.  So when torch is already imported -- the multi-GPU
    bench, the on-device algebra of model.gaussian_process -- let it go
    first.  Nothing is imported here: the solver itself never needs torch."""
    import sys
    torch = sys.modules.get('torch')
    if torch is not None:
        try:
            torch.cuda.is_available()
        except Exception:
            pass


def lib():
    """The loaded library with argtypes set (builds it on first use)."""
    global _lib
    with _lock:
        if _lib is None:
            _let_torch_initialise_first()
            L = ctypes.CDLL(build_library())
            for name, argtypes in SIGNATURES.items():
                fn = getattr(L, name)
                fn.argtypes = argtypes
                fn.restype = ctypes.c_int
            L.gd_last_error.restype = ctypes.c_char_p
            L.gd_last_error.argtypes = []
            L.gd_version.restype = ctypes.c_char_p
            L.gd_version.argtypes = []
            _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise HIPError(lib().gd_last_error().decode(errors='replace'))


def ensure_device(device=None):
    """Select the HIP device of this process (honours LOCAL_RANK for
    one-process-per-GPU launches) and make it current for the calling
    thread.  The choice is process-wide, but hipSetDevice is per host thread:
    a kernel evaluation from a worker thread (thread pool, joblib threading)
    would otherwise run with device 0 current while its buffers, modules and
    streams belong to device LOCAL_RANK."""
    global _device
    if _device is None or (device is not None and device != _device):
        if device is None:
            device = int(os.environ.get('GD_DEVICE',
                                        os.environ.get('LOCAL_RANK', 0)))
        n = ctypes.c_int(0)
        check(lib().gd_device_count(ctypes.byref(n)))
        if n.value <= 0:
            raise HIPError('no HIP device')
        if _device != device % n.value:
            first = _device is None
            check(lib().gd_init(device % n.value))
            _device = device % n.value
            _thread.device = _device
            if first and os.environ.get('GD_PINNED_RESULTS', '1') != '0':
                # (with the context: two result blocks pinned ahead of the
                # first evaluation, 3 ms once per process)
                warm_pinned_pool()
    if getattr(_thread, 'device', None) != _device:
        check(lib().gd_set_device(_device))
        _thread.device = _device
    return _device


def device_props(device=None):
    dev = ensure_device(device)
    p = DeviceProps()
    check(lib().gd_device_props(dev, ctypes.byref(p)))
    return p


STAGED_UPLOAD_BYTES = 256 << 10
STAGING_CHUNK_BYTES = 64 << 20      # larger transfers go in pieces of this size
_staging_lock = threading.Lock()
_staging_ptr, _staging_view = None, None


def _staging(nbytes):
    """uint8 view of the pinned staging buffer, grown to at least `nbytes`
    (hipHostMalloc: 0.7 ms for 4 MB, once).  Call with _staging_lock held."""
    global _staging_ptr, _staging_view
    import numpy as np
    if _staging_view is None or len(_staging_view) < nbytes:
        if _staging_ptr is not None:
            _staging_view = None
            lib().gd_host_free(_staging_ptr)
            _staging_ptr = None
        cap = max(int(nbytes), 8 << 20)
        p = ctypes.c_void_p()
        check(lib().gd_host_alloc(ctypes.byref(p), cap))
        _staging_ptr = p.value
        _staging_view = np.frombuffer(
            (ctypes.c_uint8 * cap).from_address(p.value), dtype=np.uint8)
    return _staging_view


# --------------------------------------------------------------------------
# pinned host arrays for results
# --------------------------------------------------------------------------
#: result arrays of this many bytes and more come from the pinned pool
PINNED_MIN_BYTES = 256 << 10
#: ... up to this size each (a nodal matrix of gigabytes stays pageable)
PINNED_MAX_BYTES = 96 << 20
#: pinned bytes the pool may hold idle (what is lent out is not counted)
PINNED_POOL_BYTES = 256 << 20
#: pinned bytes that may be LENT OUT at a time (GD_PINNED_LIMIT_MB): a caller
#: who keeps many result arrays -- a hyperparameter sweep storing every Gram
#: matrix -- would otherwise pin gigabytes of host memory that cannot be
#: swapped; past the limit results are ordinary pageable arrays (they take
#: the staging copy, 0.7 ms per 8 MB)
PINNED_LIMIT_BYTES = int(os.environ.get('GD_PINNED_LIMIT_MB', 1024)) << 20
_pinned_lock = threading.Lock()
_pinned_free = {}        # capacity -> [pointer, ...]
_pinned_idle = 0
_pinned_live = {}        # pointer -> capacity (lent out)
_pinned_lent = 0         # sum of the capacities lent out


class _PinnedBlock:
    """Owner of one pinned allocation lent to a numpy array: numpy keeps it
    as the `base` of the array and of every view of it; when the last one is
    gone the block goes back to the pool."""

    def __init__(self, ptr, cap, nbytes):
        self.ptr, self.cap, self.pid = ptr, cap, os.getpid()
        self.__array_interface__ = dict(
            shape=(nbytes,), typestr='|u1', data=(ptr, False), version=3)

    def __del__(self):
        global _pinned_idle, _pinned_lent
        # (a forked child inherits the arrays, not the HIP context that pinned
        # them: the block is the parent's to free or to pool)
        if _lib is None or self.pid != os.getpid():
            return
        with _pinned_lock:
            if _pinned_live.pop(self.ptr, None) is not None:
                _pinned_lent -= self.cap
            if _pinned_idle + self.cap <= PINNED_POOL_BYTES:
                _pinned_free.setdefault(self.cap, []).append(self.ptr)
                _pinned_idle += self.cap
                return
        try:
            _lib.gd_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(size, dtype):
    """A 1-D numpy array of `size` elements in pinned host memory
    (hipHostMalloc through gd_host_alloc), from a process-wide pool: the
    device copies results into it by DMA, where a pageable array takes the
    bytes through the pinned staging buffer and one more memcpy (0.7 ms for
    the 8 MB matrix of 1000 graphs, 5 ms for its seven gradient planes).  The
    role of the reference's managed-memory output arrays
    (graphdot/cuda/array.py:14-31, `umempty`).  Blocks return to the pool when
    the array and all its views are gone; pinning costs 0.2 ms per MB, which
    is why they are kept.  Small and very large arrays stay pageable."""
    import numpy as np
    global _pinned_idle, _pinned_lent
    dtype = np.dtype(dtype)
    nbytes = int(size) * dtype.itemsize
    if not (PINNED_MIN_BYTES <= nbytes <= PINNED_MAX_BYTES) \
            or dtype.hasobject \
            or os.environ.get('GD_PINNED_RESULTS', '1') == '0':
        return np.empty(int(size), dtype)
    cap = 1 << (nbytes - 1).bit_length() if nbytes <= (8 << 20) \
        else -(-nbytes // (8 << 20)) * (8 << 20)
    ptr = None
    with _pinned_lock:
        if _pinned_lent + cap > PINNED_LIMIT_BYTES:
            return np.empty(int(size), dtype)
        _pinned_lent += cap          # (reserved; given back on failure)
        free = _pinned_free.get(cap)
        if free:
            ptr = free.pop()
            _pinned_idle -= cap
    if ptr is None:
        try:
            ensure_device()
            p = ctypes.c_void_p()
            check(lib().gd_host_alloc(ctypes.byref(p), cap))
        except HIPError:
            # (no device, or no pinnable memory left: a pageable array; the
            # evaluation that follows reports a missing device itself)
            with _pinned_lock:
                _pinned_lent -= cap
            return np.empty(int(size), dtype)
        ptr = p.value
    with _pinned_lock:
        _pinned_live[ptr] = cap
    block = _PinnedBlock(ptr, cap, nbytes)
    return np.asarray(block).view(dtype)


def is_pinned(array):
    """Does `array` lie in a block of the pinned pool?"""
    if not _pinned_live:
        return False
    a = array.ctypes.data
    with _pinned_lock:
        for ptr, cap in _pinned_live.items():
            if ptr <= a and a + array.nbytes <= ptr + cap:
                return True
    return False


def warm_pinned_pool(sizes=(8 << 20, 8 << 20)):
    """Pin a few result blocks ahead of the first evaluation (called with
    the device context by HIPBackend: process start-up, not per call)."""
    import numpy as np
    held = [pinned_empty(n, np.uint8) for n in sizes]
    del held


class DeviceBuffer:
    """Owning handle of one hipMalloc allocation."""

    def __init__(self, nbytes):
        ensure_device()
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(lib().gd_malloc(ctypes.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value or 0

    def upload(self, array, offset=0, stream=None):
        import numpy as np
        a = np.ascontiguousarray(array)
        assert offset + a.nbytes <= self.nbytes
        if a.nbytes >= STAGED_UPLOAD_BYTES:
            # large pageable sources: through the process's pinned staging
            # buffer (a copy at memory speed + a DMA at link speed, 0.2 ms
            # for 4 MB).  Handed over directly, the runtime pins the pages of
            # a large source in place, which took 2-20 ms per array on the
            # first call of a new layout (scripts/upload_bench.py).
            src = a.reshape(-1).view(np.uint8)
            with _staging_lock:
                view = _staging(min(a.nbytes, STAGING_CHUNK_BYTES))
                for at in range(0, a.nbytes, STAGING_CHUNK_BYTES):
                    n = min(STAGING_CHUNK_BYTES, a.nbytes - at)
                    np.copyto(view[:n], src[at:at + n])
                    check(lib().gd_memcpy_h2d(self.ptr + offset + at,
                                              view.ctypes.data, n, stream))
                    # (the staging buffer is reused by the next piece)
                    check(lib().gd_stream_sync(stream))
            return self
        # pageable source: hipMemcpyAsync returns once `a` may be reused
        check(lib().gd_memcpy_h2d(self.ptr + offset, a.ctypes.data, a.nbytes,
                                  stream))
        return self

    def download(self, array, offset=0, stream=None):
        import numpy as np
        assert array.flags['C_CONTIGUOUS']
        assert offset + array.nbytes <= self.nbytes
        if array.nbytes >= STAGED_UPLOAD_BYTES and not is_pinned(array):
            # large pageable destinations: through the pinned staging buffer
            # (the 8 MB matrix of 1000 graphs into a fresh numpy array took
            # 10 ms directly: the runtime pins the untouched pages)
            dst = array.reshape(-1).view(np.uint8)
            with _staging_lock:
                view = _staging(min(array.nbytes, STAGING_CHUNK_BYTES))
                for at in range(0, array.nbytes, STAGING_CHUNK_BYTES):
                    n = min(STAGING_CHUNK_BYTES, array.nbytes - at)
                    check(lib().gd_memcpy_d2h(view.ctypes.data,
                                              self.ptr + offset + at, n,
                                              stream))
                    check(lib().gd_stream_sync(stream))
                    np.copyto(dst[at:at + n], view[:n])
            return array
        check(lib().gd_memcpy_d2h(array.ctypes.data, self.ptr + offset,
                                  array.nbytes, stream))
        check(lib().gd_stream_sync(stream))
        return array

    def zero(self, stream=None):
        check(lib().gd_memset(self.ptr, 0, self.nbytes, stream))

    def free(self):
        if getattr(self, 'ptr', 0) and _lib is not None:
            try:
                _lib.gd_free(self.ptr)
            except Exception:
                pass
            self.ptr = 0

    def __del__(self):
        self.free()


class DeviceArray:
    """A strided view of device memory that other libraries can adopt
    without a copy (``__cuda_array_interface__`` version 2; on ROCm builds of
    PyTorch ``torch.as_tensor(view, device='cuda')`` takes it).  `owner`
    keeps the allocation alive as long as the view object lives; the *content*
    is only valid until whoever owns the buffer writes it again."""

    def __init__(self, ptr, shape, dtype, strides=None, owner=None):
        import numpy as np
        dtype = np.dtype(dtype)
        self.ptr, self.shape, self.dtype, self.owner = ptr, tuple(shape), \
            dtype, owner
        self.strides = tuple(strides) if strides is not None else None
        self.__cuda_array_interface__ = dict(
            shape=self.shape, strides=self.strides, typestr=dtype.str,
            data=(int(ptr), False), version=2)

    @classmethod
    def fortran(cls, ptr, shape, dtype, owner=None):
        """Column-major view (the layout of the solver's outputs)."""
        import numpy as np
        item = np.dtype(dtype).itemsize
        strides, step = [], item
        for n in shape:
            strides.append(step)
            step *= int(n)
        return cls(ptr, shape, dtype, strides, owner)


class Event:
    def __init__(self):
        p = ctypes.c_void_p()
        check(lib().gd_event_create(ctypes.byref(p)))
        self.h = p.value

    def record(self, stream=None):
        check(lib().gd_event_record(self.h, stream))

    def sync(self):
        check(lib().gd_event_sync(self.h))

    def elapsed_ms(self, stop):
        ms = ctypes.c_float(0)
        check(lib().gd_event_elapsed_ms(self.h, stop.h, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            try:
                _lib.gd_event_destroy(self.h)
            except Exception:
                pass
            self.h = None


class Stream:
    """A non-blocking HIP stream."""

    def __init__(self, low_priority=False):
        ensure_device()
        p = ctypes.c_void_p()
        create = lib().gd_stream_create_low_priority if low_priority \
            else lib().gd_stream_create
        check(create(ctypes.byref(p)))
        self.h = p.value

    def sync(self):
        check(lib().gd_stream_sync(self.h))

    def wait_event(self, event):
        """Device-side: later work on this stream waits for `event`."""
        check(lib().gd_stream_wait_event(self.h, event.h))

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            try:
                _lib.gd_stream_destroy(self.h)
            except Exception:
                pass
            self.h = None


class Module:
    """A loaded gfx950 code object and its kernels."""

    def __init__(self, image: bytes):
        ensure_device()
        self._image = ctypes.create_string_buffer(image, len(image))
        m = ctypes.c_void_p()
        check(lib().gd_module_load(self._image, len(image), ctypes.byref(m)))
        self.h = m.value
        self._functions = {}

    def function(self, name):
        if name not in self._functions:
            f = ctypes.c_void_p()
            check(lib().gd_module_get_function(self.h, name.encode(),
                                               ctypes.byref(f)))
            self._functions[name] = f.value
        return self._functions[name]

    def attributes(self, name):
        lds, thr, regs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().gd_function_attributes(self.function(name),
                                           ctypes.byref(lds),
                                           ctypes.byref(thr),
                                           ctypes.byref(regs)))
        return dict(static_lds=lds.value, max_threads=thr.value,
                    num_regs=regs.value)


def launch(function, grid, block, args: bytes, stream=None, dynamic_lds=0,
           cooperative=False):
    """`cooperative`: every workgroup of the grid resident at once
    (gd_launch_cooperative): the kernel synchronises its workgroups through
    device memory."""
    buf = ctypes.create_string_buffer(args, len(args))
    fn = lib().gd_launch_cooperative if cooperative else lib().gd_launch
    check(fn(function, grid, block, dynamic_lds, stream, buf, len(args)))


def max_active_blocks(function, block, dynamic_lds=0):
    """Workgroups of `block` threads and `dynamic_lds` bytes that one
    compute unit holds at a time."""
    n = ctypes.c_int(0)
    check(lib().gd_function_max_active_blocks(function, block, dynamic_lds,
                                              ctypes.byref(n)))
    return n.value


def null_stream_wait_event(event):
    check(lib().gd_stream_wait_event(None, event.h))


def set_max_dynamic_lds(function, nbytes):
    """Best effort: module-loaded kernels on ROCm accept large dynamic LDS
    requests directly; a refusal here is not fatal (the launch itself reports
    an over-size request)."""
    return lib().gd_function_set_max_dynamic_lds(function, int(nbytes)) == 0


def synchronize(stream=None):
    if stream is None:
        check(lib().gd_device_sync())
    else:
        check(lib().gd_stream_sync(stream))


class Communicator:
    """RCCL communicator of this process (one process per GPU), bound through
    the C ABI (`gd_comm_*`, include/gdhip.h).  The 128-byte unique id is made
    by rank 0 (`Communicator.unique_id()`) and handed to every rank out of
    band; `all_gather` is asynchronous on the given stream."""
    DTYPES = {'float32': 0, 'float64': 1, 'uint8': 2}

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib().gd_comm_unique_id(buf))
        return buf.raw

    def __init__(self, n_ranks, unique_id, rank):
        ensure_device()
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        c = ctypes.c_void_p()
        check(lib().gd_comm_init_rank(ctypes.byref(c), self.n_ranks,
                                      ctypes.c_char_p(unique_id), self.rank))
        self.h = c.value

    def all_gather(self, send_ptr, recv_ptr, count, dtype, stream=None):
        import numpy as np
        check(lib().gd_all_gather(send_ptr, recv_ptr, int(count),
                                  self.DTYPES[np.dtype(dtype).name], self.h,
                                  stream))

    def destroy(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.gd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
