#!/usr/bin/env python3
"""What does the first device-to-host copy of a process pay, and does a small
warm-up copy pay it?   python scripts/download_bench2.py [warm bytes]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from graphdot_amd.hip import runtime
runtime.ensure_device()
L = runtime.lib()
n = 8 << 20
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
buf = runtime.DeviceBuffer(n)
buf.zero(); runtime.synchronize()
view = runtime._staging(n)
def d2h(nbytes):
    t0 = time.perf_counter()
    runtime.check(L.gd_memcpy_d2h(view.ctypes.data, buf.ptr, nbytes, None))
    runtime.check(L.gd_stream_sync(None))
    return 1e3 * (time.perf_counter() - t0)
if warm:
    print(f'warm-up copy of {warm} B: {d2h(warm):.2f} ms')
print(f'first 8 MB copy: {d2h(n):.2f} ms, second {d2h(n):.2f} ms')
