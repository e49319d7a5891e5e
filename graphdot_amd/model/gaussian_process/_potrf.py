"""Host side of the blocked Cholesky factorisation of potrf.hip: compiles the
three kernels once (JIT cache of graphdot_amd.hip.jit, IEEE arithmetic: no
fast-math) and runs them on a float64 torch tensor in place.  The launches go
to torch's *current* stream of the tensor's device, so that they are ordered
against the torch operations around them (the clone before, tril / the
triangular solve after) whatever stream the caller works on."""
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
            _kernels = (mod, mod.function('potrf_diag_f64'),
                        mod.function('potrf_panel_f64'),
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
    _, diag, panel, syrk = _load()
    ptr = A.data_ptr()
    with torch.cuda.device(A.device):
        stream = torch.cuda.current_stream().cuda_stream or None
        # L_kk^-1 of the panel in flight (64 x 64): written by the one
        # workgroup that factors the diagonal block, read by the panel launch
        # behind it (the caching allocator keeps the block tied to this stream)
        work = torch.empty(_B * _B, dtype=torch.float64, device=A.device)
    nb = -(-n // _B)

    def args(k0):
        return struct.pack('<QiiiiQ', ptr, ld, n, k0, 0, work.data_ptr())  # (pad: Linv is 8-aligned)
    runtime.launch(diag, 1, 256, args(0), stream=stream)
    for kb in range(nb - 1):
        a = args(kb * _B)
        m = nb - kb - 1              # row blocks below the diagonal block
        runtime.launch(panel, m, 256, a, stream=stream)
        # (tile 0 of the trailing update is the next diagonal block: its
        # workgroup factors it and refills `work`)
        runtime.launch(syrk, m * (m + 1) // 2, 256, a, stream=stream)
    return A
