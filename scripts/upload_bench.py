#!/usr/bin/env python3
"""Host-to-device copy of a job list: pageable numpy array against a pinned
staging buffer (what the first call of a new layout pays)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from graphdot_amd.hip import runtime
runtime.ensure_device()
L = runtime.lib()
for mb in (1, 2, 4, 8):
    n = mb << 20
    t0 = time.perf_counter(); buf = runtime.DeviceBuffer(n); t_alloc = time.perf_counter() - t0
    a = np.random.randint(0, 255, n, dtype=np.uint8)          # touched pages
    t0 = time.perf_counter(); buf.upload(a); runtime.synchronize(); t_page = time.perf_counter() - t0
    b = np.random.randint(0, 255, n, dtype=np.uint8)
    t0 = time.perf_counter(); buf.upload(b); runtime.synchronize(); t_page2 = time.perf_counter() - t0
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); runtime.check(L.gd_host_alloc(ctypes.byref(p), n)); t_pin = time.perf_counter() - t0
    v = np.frombuffer((ctypes.c_uint8 * n).from_address(p.value), dtype=np.uint8)
    t0 = time.perf_counter(); np.copyto(v, a); t_copy = time.perf_counter() - t0
    t0 = time.perf_counter(); runtime.check(L.gd_memcpy_h2d(buf.ptr, p.value, n, None)); runtime.synchronize(); t_h2d = time.perf_counter() - t0
    t0 = time.perf_counter(); np.copyto(v, b); runtime.check(L.gd_memcpy_h2d(buf.ptr, p.value, n, None)); runtime.synchronize(); t_both = time.perf_counter() - t0
    print(f'{mb} MB: hipMalloc {1e3*t_alloc:.2f} ms  pageable upload {1e3*t_page:.2f} / {1e3*t_page2:.2f} ms  '
          f'hipHostMalloc {1e3*t_pin:.2f} ms  copy into pinned {1e3*t_copy:.2f} ms  pinned upload {1e3*t_h2d:.2f} ms  copy+upload {1e3*t_both:.2f} ms')
    L.gd_host_free(p)
