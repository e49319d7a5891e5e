"""Host side of the blocked Cholesky factorisation of potrf.hip: compiles the
two kernels once (JIT cache of graphdot_amd.hip.jit, IEEE arithmetic: no
fast-math) and runs them on a float64 torch tensor in place.  The launches go
to the null stream, where torch's own work of this process is ordered too."""
import os
import struct
import threading

_SOURCE = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                       'potrf.hip')
_FLAGS = ('-fno-fast-math',)
_B = 64
_lock = threading.Lock()
_kernels = None


def source():
    with open(_SOURCE) as f:
        return f.read()


def precompile():
    """Compile into the JIT cache (hipcc, no device needed)."""
    from ...hip import jit
    return jit.compile_source(source(), _FLAGS)


def _load():
    global _kernels
    with _lock:
        if _kernels is None:
            from ...hip import jit, runtime
            mod = runtime.Module(jit.load_image(precompile()))
            _kernels = (mod, mod.function('potrf_panel_f64'),
                        mod.function('syrk_update_f64'))
    return _kernels


def cholesky_(A):
    """Lower Cholesky factor of the symmetric positive definite float64 CUDA
    tensor `A` (n x n), computed in the memory of `A`.  Returns a tensor (a
    view of `A`) with L in its lower triangle *of every diagonal block and
    below*; tiles above the diagonal keep the input -- take ``torch.tril`` of
    what is returned.  Rows or columns must be contiguous (a column-major
    matrix is its own transpose's row-major image: symmetric, so the view
    ``A.T`` is factored).  Not positive definite: NaN on the diagonal."""
    import torch
    from ...hip import runtime
    assert A.is_cuda and A.dtype == torch.float64 and A.dim() == 2
    n = A.shape[0]
    assert A.shape[1] == n
    if n > 1 and A.stride(1) != 1:
        assert A.stride(0) == 1, 'rows or columns must be contiguous'
        A = A.T
    ld = A.stride(0) if n > 1 else 1
    _, panel, syrk = _load()
    ptr = A.data_ptr()
    # (torch's current stream of a default-configured process is the null
    # stream; a caller on another stream orders against it with events)
    nb = -(-n // _B)
    for kb in range(nb):
        k0 = kb * _B
        args = struct.pack('<Qiii', ptr, ld, n, k0)
        runtime.launch(panel, nb - kb, 256, args)
        m = nb - kb - 1
        if m > 0:
            runtime.launch(syrk, m * (m + 1) // 2, 256, args)
    return A
