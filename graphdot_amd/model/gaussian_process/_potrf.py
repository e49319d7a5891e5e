"""Host side of potrf.hip, the one-launch Cholesky factorisation and inverse:
compiles the kernel once (JIT cache of graphdot_amd.hip.jit, IEEE arithmetic:
no fast-math) and runs it on float64 torch tensors.  The launch goes to
torch's *current* stream of the tensor's device, so that it is ordered
against the torch operations around it (the clone before, whatever reads the
inverse after) whatever stream the caller works on."""
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
            _kernels = (mod, mod.function('spd_factor_invert_f64'))
    return _kernels


class FactorisationError(RuntimeError):
    """The data-flow launch gave up on a wait (its status word is set)."""


def _dataflow(A, invert, stamps=None):
    """One launch of `spd_factor_invert_f64` on the row-contiguous float64
    CUDA tensor `A` (overwritten: L in its lower tiles).  Returns (Kinv or
    None, head): `head` is the int32 tensor whose first 16 + 2 nb words are
    [next role, status, pad][nb doubles: log-determinant shares of the
    diagonal blocks] -- one small download tells the caller whether the
    launch completed and what log|L| is."""
    import torch
    from ...hip import runtime
    n = A.shape[0]
    nb = -(-n // _B)
    ld = A.stride(0) if n > 1 else 1
    fn = _load()[1]
    with torch.cuda.device(A.device):
        stream = torch.cuda.current_stream().cuda_stream or None
        head = torch.zeros(16 + 2 * nb + 2 * nb * nb, dtype=torch.int32,
                           device=A.device)
        linv = torch.empty(nb * _B * _B, dtype=torch.float64, device=A.device)
        if invert:
            Kinv = torch.empty((n, n), dtype=torch.float64, device=A.device)
            Z = torch.empty(nb * nb * _B * _B, dtype=torch.float64,
                            device=A.device)
        else:
            Kinv = Z = None
    roles = nb * nb + (nb * (nb + 1) // 2 if invert else 0)
    args = struct.pack('<QQQQQiiiiQ', A.data_ptr(),
                       Kinv.data_ptr() if invert else 0,
                       Z.data_ptr() if invert else 0, linv.data_ptr(),
                       head.data_ptr(), ld, n, n, 1 if invert else 0,
                       stamps.data_ptr() if stamps is not None else 0)
    # (roles are handed out by an atomic counter in dependency order: any
    # grid size is deadlock-free; 512 = two workgroups per compute unit)
    runtime.launch(fn, min(roles, 512), 256, args, stream=stream)
    # keep the workspaces alive until the stream has passed the launch: the
    # caching allocator ties the blocks to the current stream, and every
    # later use of them is ordered behind the launch there
    return Kinv, head


def parse_head(words, nb):
    """(completed, log|L|) from the first 16 + 2 nb int32 words of the
    launch's sync buffer, given as 8 + nb float64 (how the regressor packs
    them into its one download) or as the int32 words themselves."""
    import numpy as np
    words = np.ascontiguousarray(words)
    status = int(words.view(np.int32)[1])
    shares = words.view(np.float64)[8:8 + nb]
    return status == 0, float(shares.sum())


def read_head(head, nb):
    """(completed, log|L|) from the first words of the launch's sync buffer:
    ONE device-to-host copy (it synchronises with the launch)."""
    return parse_head(head[:16 + 2 * nb].cpu().numpy(), nb)


def factor_inverse(K):
    """``K^-1`` and ``log|K|`` of the symmetric positive definite float64
    CUDA tensor `K` in ONE launch (potrf.hip, `spd_factor_invert_f64`):
    returns (Kinv, head, nb); `read_head(head, nb)` gives (completed,
    log|L|) -- ``log|K| = 2 log|L|``, NaN when `K` is not positive definite.
    `K` is left untouched (the factor is computed in a copy)."""
    import torch
    assert K.is_cuda and K.dtype == torch.float64 and K.dim() == 2
    n = K.shape[0]
    assert K.shape[1] == n and n >= 1
    A = K.clone(memory_format=torch.contiguous_format)
    Kinv, head = _dataflow(A, True)
    return Kinv, head, -(-n // _B)


def cholesky_(A):
    """Lower Cholesky factor of the symmetric positive definite float64 CUDA
    tensor `A` (n x n), computed in the memory of `A` (the same launch without
    the roles of the inverse).  Returns a tensor (a view of `A`) with L in its
    lower triangle *of every diagonal block and below*; tiles above the
    diagonal keep the input -- take ``torch.tril`` of what is returned.  Rows
    or columns must be contiguous (a column-major matrix is its own
    transpose's row-major image: symmetric, so the view ``A.T`` is factored).
    Not positive definite: NaN on the diagonal."""
    import torch
    assert A.is_cuda and A.dtype == torch.float64 and A.dim() == 2
    n = A.shape[0]
    assert A.shape[1] == n
    if n > 1 and A.stride(1) != 1:
        assert A.stride(0) == 1, 'rows or columns must be contiguous'
        A = A.T
    _, head = _dataflow(A, False)
    A._gd_head = head          # (the launch's status word, for the tests)
    return A
