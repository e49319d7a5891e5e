#!/usr/bin/env python3
"""The blocked Cholesky of potrf.hip against torch.linalg.cholesky (float64,
MI355X), and the whole factor step of the GPR (factor + inverse + log-det)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from graphdot_amd.model.gaussian_process._potrf import cholesky_
from graphdot_amd.model.gaussian_process.gpr import _Dense


def timed(f, reps=30):
    for _ in range(3):
        r = f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, r


for n in (250, 500, 1000, 2000, 4000):
    A = torch.randn(n, n, dtype=torch.float64, device='cuda')
    K = A @ A.T / n + torch.eye(n, dtype=torch.float64, device='cuda')
    t_lib, ref = timed(lambda: torch.linalg.cholesky(K))
    t_own, L = timed(lambda: torch.tril(cholesky_(K.clone())))
    err = float((L - ref).abs().max() / ref.abs().max())
    la = _Dense('cuda')
    la.native_cholesky = True
    t_f1, _ = timed(lambda: la.factor(K, 1e-8))
    la.native_cholesky = False
    t_f0, _ = timed(lambda: la.factor(K, 1e-8))
    print(f'n={n}: cholesky library {t_lib:.3f} ms, potrf.hip {t_own:.3f} ms '
          f'(max rel diff {err:.1e}); factor step {t_f0:.3f} -> {t_f1:.3f} ms')
