#!/usr/bin/env python3
"""Stress of the one-launch factor-and-invert (potrf.hip): random sizes,
conditioning and strides, with and without a second stream keeping the compute
units busy, every word of the inverse and of the factor against the library.
Hand-offs between workgroups that go wrong (a stale tile, a flag ahead of its
stores) show up as wrong numbers here, not as hangs.
    python scripts/potrf_stress.py [rounds] [--seed=N]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np                                              # noqa: E402
import torch                                                    # noqa: E402
from graphdot_amd.model.gaussian_process._potrf import (        # noqa: E402
    cholesky_, factor_inverse, read_head)

rounds = int(next((a for a in sys.argv[1:] if not a.startswith('--')), 60))
seed = int(next((a.split('=')[1] for a in sys.argv[1:]
                 if a.startswith('--seed=')), 0))
rng = np.random.default_rng(seed)
g = torch.Generator(device='cuda').manual_seed(seed)
hog, side = torch.cuda.Stream(), torch.cuda.Stream()
B1 = torch.randn(3072, 3072, device='cuda')
worst = dict(inv=0.0, chol=0.0, logdet=0.0)
for it in range(rounds):
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 1200),
                        rng.integers(1200, 2600)], p=[0.3, 0.5, 0.2]))
    A = torch.randn(n, n, dtype=torch.float64, device='cuda', generator=g)
    ridge = float(rng.choice([1e-3, 1e-1, 1.0]))
    K = A @ A.T / n + ridge * torch.eye(n, dtype=torch.float64, device='cuda')
    if rng.integers(2):                      # column-major, as device_gram hands it over
        K = K.T.contiguous().T
    busy = bool(rng.integers(2))
    torch.cuda.synchronize()
    if busy:
        with torch.cuda.stream(hog):
            for _ in range(20):
                B1 @ B1
    with torch.cuda.stream(side):
        Kinv, head, nb = factor_inverse(K)
        L = torch.tril(cholesky_(K.clone()))
    torch.cuda.synchronize()
    ok, ld = read_head(head, nb)
    assert ok, ('poisoned', it, n)
    ref = torch.linalg.inv(K)
    e_inv = float((Kinv - ref).abs().max() / ref.abs().max())
    e_chol = float((L - torch.linalg.cholesky(K)).abs().max())
    e_ld = abs(2 * ld - float(torch.logdet(K))) / max(1.0, abs(float(torch.logdet(K))))
    cond = float(torch.linalg.cond(K)) if n <= 600 else float('nan')
    # (the inverse is good to cond(K) eps like the library's)
    tol = 1e-11 if ridge >= 1e-1 else 1e-9
    assert e_inv < tol and e_chol < 1e-11 and e_ld < 1e-11, \
        (it, n, ridge, busy, e_inv, e_chol, e_ld, cond)
    for k_, v_ in (('inv', e_inv), ('chol', e_chol), ('logdet', e_ld)):
        worst[k_] = max(worst[k_], v_)
print(f'potrf stress ok: {rounds} rounds, seed {seed}, worst {worst}')
