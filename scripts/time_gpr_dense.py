#!/usr/bin/env python3
"""The dense algebra of one GPR likelihood + gradient step, piece by piece
(n = 1000, n_theta gradient planes, float64, MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import numpy as np
from graphdot_amd.model.gaussian_process.gpr import _Dense
from graphdot_amd.model.gaussian_process._potrf import cholesky_
n, nt = 1000, 6
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
A = torch.randn(n, n, generator=g, dtype=torch.float64)
K = (A @ A.T / n + torch.eye(n, dtype=torch.float64)).to(dev)
dK = torch.randn(nt, n, n, generator=g, dtype=torch.float64).to(dev).permute(1, 2, 0)   # planes contiguous
y = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
la = _Dense('cuda')
def timed(name, f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    print(f'{name:44s} {1e3 * (time.perf_counter() - t0) / reps:7.3f} ms')
    return r
timed('clone + potrf.hip + tril', lambda: torch.tril(cholesky_(K.clone())))
L = torch.tril(cholesky_(K.clone()))
timed('finite / positive check (host sync)', lambda: bool((torch.isfinite(torch.diagonal(L)) & (torch.diagonal(L) > 0)).all()))
I = torch.eye(n, dtype=torch.float64, device=dev)
X = timed('solve_triangular(L, I)', lambda: torch.linalg.solve_triangular(L, I, upper=False))
Kinv = timed('X^T X', lambda: X.T @ X)
timed('eye(n)', lambda: torch.eye(n, dtype=torch.float64, device=dev))
timed('logdet (host sync)', lambda: float(2.0 * torch.log(torch.diagonal(L)).sum()))
Ky = timed('Kinv @ y', lambda: Kinv @ y)
timed('y @ Ky (host sync)', lambda: float(y @ Ky))
timed('(Kinv[..., None] * dK).sum((0, 1))', lambda: (Kinv.unsqueeze(-1) * dK).sum((0, 1)))
timed('Ky @ tensordot(Ky, dK)', lambda: Ky @ torch.tensordot(Ky, dK, dims=([0], [0])))
W = Kinv - torch.outer(Ky, Ky)
timed('W = Kinv - Ky Ky^T', lambda: Kinv - torch.outer(Ky, Ky))
timed('(W[..., None] * dK).sum((0, 1))', lambda: (W.unsqueeze(-1) * dK).sum((0, 1)))
dKp = dK.permute(2, 0, 1)
timed('(dKp * W).sum((1, 2))  [plane-major]', lambda: (dKp * W).sum((1, 2)))
timed('dKp.reshape(nt, -1) @ W.reshape(-1)', lambda: dKp.reshape(nt, -1) @ W.reshape(-1))
timed('factor() as a whole', lambda: la.factor(K, 1e-10))
timed('.cpu() of the gradient', lambda: (dKp.reshape(nt, -1) @ W.reshape(-1)).cpu())
