"""Round 6: the one-launch factor-and-invert of potrf.hip
(`spd_factor_invert_f64`) against the library: values for sizes around the
tile edge, timings at the benchmark sizes.  Usage: python scripts/time_spd_inverse.py [--sizes 500,1000,2000]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                   # noqa: E402
from graphdot_amd.model.gaussian_process import _potrf         # noqa: E402


def spd(n, g):
    A = torch.randn(n, n, dtype=torch.float64, device='cuda', generator=g)
    return A @ A.T / n + 0.1 * torch.eye(n, dtype=torch.float64, device='cuda')


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def factor_then_library(K):
    """Round 5's dense half with this launch as its factorisation: L, then
    L^-1 by a triangular solve and K^-1 = L^-T L^-1 by a product (the chain of
    31 launches it had for L is gone; its timings are in
    profiles/r06_spd_inverse_v1.json: 0.67 / 1.19 / 2.57 / 8.38 ms at
    n = 500 / 1000 / 2000 / 4000)."""
    L = torch.tril(_potrf.cholesky_(K.clone()))
    X = torch.linalg.solve_triangular(
        L, torch.eye(len(L), dtype=L.dtype, device=L.device), upper=False)
    return X.T @ X, L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='500,1000,2000,4000')
    ap.add_argument('--check', default='1,2,5,63,64,65,128,130,200,500,1000,1037')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    g = torch.Generator(device='cuda').manual_seed(0)
    report = {'check': [], 'time': []}
    for n in [int(x) for x in a.check.split(',') if x]:
        K = spd(n, g)
        Kinv, head, nb = _potrf.factor_inverse(K)
        ok, ld = _potrf.read_head(head, nb)
        ref = torch.linalg.inv(K)
        err = float((Kinv - ref).abs().max() / ref.abs().max())
        resid = float((Kinv @ K - torch.eye(n, dtype=torch.float64,
                                            device='cuda')).abs().max())
        ldref = float(torch.logdet(K))
        Lf = torch.tril(_potrf.cholesky_(K.clone()))
        lerr = float((Lf - torch.linalg.cholesky(K)).abs().max())
        report['check'].append(dict(n=n, completed=ok, inv_err=err,
                                    resid=resid, logdet_err=abs(2 * ld - ldref),
                                    chol_err=lerr))
        print(report['check'][-1], flush=True)
    for n in [int(x) for x in a.sizes.split(',') if x]:
        K = spd(n, g)
        t_new = timed(lambda: _potrf.factor_inverse(K))
        t_chol = timed(lambda: _potrf.cholesky_(K.clone()))
        t_old = timed(lambda: factor_then_library(K))
        t_lib = timed(lambda: torch.cholesky_inverse(torch.linalg.cholesky(K)),
                      reps=5)
        t0 = time.perf_counter()
        for _ in range(20):
            Kinv, head, nb = _potrf.factor_inverse(K)
            _potrf.read_head(head, nb)
        t_sync = (time.perf_counter() - t0) / 20 * 1e3
        report['time'].append(dict(n=n, one_launch_ms=t_new,
                                   one_launch_factor_only_ms=t_chol,
                                   factor_only_then_library_inverse_ms=t_old, library_ms=t_lib,
                                   one_launch_with_download_ms=t_sync))
        print(report['time'][-1], flush=True)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(report, f, indent=1)


if __name__ == '__main__':
    main()
