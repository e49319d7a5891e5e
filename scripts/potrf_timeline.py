"""Timeline of the critical path of `spd_factor_invert_f64` (potrf.hip) from
its in-kernel wall-clock stamps (100 MHz, the same clock on every compute
unit): per panel j -- factor of A_jj, hand-over of L_jj^-1 to the role of
the next diagonal block, which makes L_j+1,j and its share of A_j+1,j+1
itself, the next factor.
Usage: python scripts/potrf_timeline.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402
from graphdot_amd.model.gaussian_process import _potrf         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator(device='cuda').manual_seed(0)
A = torch.randn(n, n, dtype=torch.float64, device='cuda', generator=g)
K = A @ A.T / n + 0.1 * torch.eye(n, dtype=torch.float64, device='cuda')
nb = -(-n // 64)
roles = nb * nb + nb * (nb + 1) // 2
for rep in range(3):
    stamps = torch.zeros(roles * 16, dtype=torch.int64, device='cuda')
    _potrf._dataflow(K.clone(), True, stamps=stamps)
    torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(roles, 16).astype(np.float64) * 0.01   # us
t0 = t[t > 0].min()
t = np.where(t > 0, t - t0, np.nan)


def role_a(i, j):
    return j * nb + (i - j)


print(f'n = {n}: first stamp to last stamp {np.nanmax(t):.1f} us')
print('panel | start | L^-1 of the block above: seen, staged | both products '
      'done | factor: begin end | L^-1 published   (all us from launch)')
for j in range(nb):
    d = t[role_a(j, j)]
    print(f'{j:3d} | {d[0]:7.1f} | {d[1]:7.1f} {d[2]:7.1f} | {d[3]:7.1f} | '
          f'{d[4]:7.1f} {d[5]:7.1f} | {d[6]:7.1f}')
dd = np.array([t[role_a(j, j)] for j in range(nb)])
print('means over the panels (us):')
print(f'  factor_block                              {np.nanmean(dd[:, 5] - dd[:, 4]):6.2f}')
print(f'  scale + store L_jj^-1 + publish           {np.nanmean(dd[:, 6] - dd[:, 5]):6.2f}')
print(f'  publish -> flag seen by the next diagonal {np.nanmean(dd[1:, 1] - dd[:-1, 6]):6.2f}')
print(f'  stage L_jj^-1                             {np.nanmean(dd[1:, 2] - dd[1:, 1]):6.2f}')
print(f'  L_j+1,j = . L_jj^-T, store, S += L L^T    {np.nanmean(dd[1:, 3] - dd[1:, 2]):6.2f}')
print(f'  to the factor layout + publish L_j+1,j    {np.nanmean(dd[1:, 4] - dd[1:, 3]):6.2f}')
print(f'  panel period                              {np.nanmean(np.diff(dd[:, 6])):6.2f}')
