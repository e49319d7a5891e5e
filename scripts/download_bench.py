#!/usr/bin/env python3
"""Device-to-host copy of an 8 MB result into a fresh numpy array: direct
(pageable destination) against the pinned staging buffer, piece by piece."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from graphdot_amd.hip import runtime
runtime.ensure_device()
L = runtime.lib()
n = 8 << 20
buf = runtime.DeviceBuffer(n)
buf.zero(); runtime.synchronize()
def t(f):
    t0 = time.perf_counter(); r = f(); return 1e3 * (time.perf_counter() - t0), r
for trial in range(3):
    ms_alloc, out = t(lambda: np.empty(n, np.uint8))
    ms_direct, _ = t(lambda: (runtime.check(L.gd_memcpy_d2h(out.ctypes.data, buf.ptr, n, None)), runtime.check(L.gd_stream_sync(None))))
    ms_again, _ = t(lambda: (runtime.check(L.gd_memcpy_d2h(out.ctypes.data, buf.ptr, n, None)), runtime.check(L.gd_stream_sync(None))))
    out2 = np.empty(n, np.uint8)
    ms_stage_alloc, view = t(lambda: runtime._staging(n))
    ms_d2h, _ = t(lambda: (runtime.check(L.gd_memcpy_d2h(view.ctypes.data, buf.ptr, n, None)), runtime.check(L.gd_stream_sync(None))))
    ms_copy, _ = t(lambda: np.copyto(out2, view[:n]))
    out3 = np.empty(n, np.uint8)
    ms_touch, _ = t(lambda: out3.fill(0))
    ms_direct_touched, _ = t(lambda: (runtime.check(L.gd_memcpy_d2h(out3.ctypes.data, buf.ptr, n, None)), runtime.check(L.gd_stream_sync(None))))
    print(f'trial {trial}: np.empty {ms_alloc:.2f}  direct into fresh {ms_direct:.2f}  again {ms_again:.2f} | '
          f'staging alloc {ms_stage_alloc:.2f}  d2h pinned {ms_d2h:.2f}  copy to fresh {ms_copy:.2f} | '
          f'touch {ms_touch:.2f}  direct into touched {ms_direct_touched:.2f}')
