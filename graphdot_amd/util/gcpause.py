"""Pausing the cyclic garbage collector for the host side of a backend call.

The host half of a first call makes a few thousand small objects (packed-graph
handles, cookies), none of them part of a cycle; if the collector's threshold
falls inside, a generation-2 pass walks the CALLER's heap -- a thousand graph
objects with their frames -- for ~8 ms of a 13 ms call.  `paused()` keeps the
collector off for the duration of the call.

The collector switch is process-wide, so the pause is counted: the first
thread to enter switches the collector off (if it was on), the last one to
leave switches it back on -- a second thread's exit never re-enables it under
a thread that is still inside.  `GD_PAUSE_GC=0` turns the pause off
(INTEGRATION.md section 3a)."""
import gc
import os
import threading
from contextlib import contextmanager

_lock = threading.Lock()
_depth = 0
_restore = False
ENABLED = os.environ.get('GD_PAUSE_GC', '1') != '0'


@contextmanager
def paused():
    global _depth, _restore
    if not ENABLED:
        yield
        return
    with _lock:
        if _depth == 0:
            _restore = gc.isenabled()
            if _restore:
                gc.disable()
        _depth += 1
    try:
        yield
    finally:
        with _lock:
            _depth -= 1
            if _depth == 0 and _restore:
                gc.enable()
