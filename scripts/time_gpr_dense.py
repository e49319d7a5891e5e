#!/usr/bin/env python3
"""The dense algebra of one GPR likelihood + gradient step, piece by piece
(n = 1000, n_theta gradient planes, float64, MI355X): the round-6 path --
one-launch factor-and-invert (potrf.hip), K^-1 y, W, the contraction with the
gradient planes, ONE download."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import numpy as np
from graphdot_amd.model.gaussian_process.gpr import _contract_planes
from graphdot_amd.model.gaussian_process._potrf import factor_inverse, parse_head
n, nt = 1000, 7
dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(0)
A = torch.randn(n, n, generator=g, dtype=torch.float64)
K = (A @ A.T / n + torch.eye(n, dtype=torch.float64)).to(dev)
dK = torch.randn(nt, n, n, generator=g, dtype=torch.float64).to(dev).permute(2, 1, 0)   # planes contiguous
y = torch.randn(n, generator=g, dtype=torch.float64).to(dev)


def timed(name, f, reps=50):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    print(f'{name:52s} {1e3 * (time.perf_counter() - t0) / reps:7.3f} ms')
    return r


timed('K.clone()', lambda: K.clone())
Kinv, head, nb = timed('factor_inverse(K) (clone + workspaces + launch)', lambda: factor_inverse(K))
Ky = timed('Kinv @ y', lambda: Kinv @ y)
timed('y @ Ky', lambda: y @ Ky)
W = timed('W = Kinv - outer(Ky, Ky)', lambda: Kinv - torch.outer(Ky, Ky))
d = timed('contraction with the planes', lambda: _contract_planes(W, dK))
timed('cat + .cpu()', lambda: torch.cat((head[:16 + 2 * nb].view(torch.float64), (y @ Ky).reshape(1), d)).cpu())


def whole():
    Kinv, head, nb = factor_inverse(K)
    Ky = Kinv @ y
    W = Kinv - torch.outer(Ky, Ky)
    return torch.cat((head[:16 + 2 * nb].view(torch.float64), (y @ Ky).reshape(1),
                      _contract_planes(W, dK))).cpu()


t0 = time.perf_counter()
for _ in range(50):
    whole()
print(f'{"the whole dense part, one download per step":52s} {1e3 * (time.perf_counter() - t0) / 50:7.3f} ms')
