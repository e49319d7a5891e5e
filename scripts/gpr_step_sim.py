#!/usr/bin/env python3
"""What ONE RANK of an N-GPU Gaussian-process likelihood step does
(configuration 5 of BASELINE.json), timed on the one GPU there is -- the
whole-step PREDICTION of DESIGN.md section 8, not a measurement of N GPUs:

  shard        value + gradient solvers of the rank's 1 / N of the pairs
  values       value solvers of the same pairs (the first step of the
               overlapped form)
  dense        the replicated part: factor and inverse (potrf.hip, one launch
               since round 6), K^-1 y, log-determinant -- on the FULL matrix
               of a previous full evaluation
  contraction  sum_p m_p W[i_p, j_p] dK[p, :] over the rank's pairs
  serial       shard, then dense, then contraction
  overlapped   values, then dense BESIDE the detached value + gradient
               solvers, then contraction (gpr.py with `overlap_min_ranks` set)

The collectives are not in it (one GPU): add the all-gather of the value
slabs, the reassembly and the all-reduce of n_theta numbers from
`bench.py --sharded`'s `phases_per_rank` (about 0.05 + 0.02 + 0.03 ms).

    python scripts/gpr_step_sim.py [--f32] [--world 8]
"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import torch                                                        # noqa: E402
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.hip import runtime                                # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import (         # noqa: E402
    HIPBackend, LaunchSet)
from graphdot_amd.kernel.marginalized._sharded import measured_shard_plan  # noqa
from graphdot_amd.model.gaussian_process.gpr import _Dense          # noqa: E402

real = np.float32 if '--f32' in sys.argv else np.float64
worlds = [int(sys.argv[sys.argv.index('--world') + 1])] \
    if '--world' in sys.argv else [2, 4, 8]
n = 1000
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
b = HIPBackend(real=real)
k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
tv, tg = k.traits(symmetric=True), k.traits(symmetric=True, eval_gradient=True)
nJ = k.n_dims
dev = torch.device('cuda', 0)

# the full matrix once: the dense part works on it
Kfull = torch.as_tensor(k.device_gram(G), device=dev).to(torch.float64).clone()
d = torch.diagonal(Kfull)
d.add_(1e-2 * float(d.mean()))
y = torch.as_tensor(np.random.default_rng(0).normal(size=n), device=dev)
la = _Dense('cuda')
ls_v, ls_g = LaunchSet(), LaunchSet()


def dense():
    Kinv, logdet = la.factor(Kfull, 1e-8)
    Ky = Kinv @ y
    return Kinv - torch.outer(Ky, Ky)


def timed(fn, steps=20):
    for _ in range(3):
        fn()
    runtime.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    runtime.synchronize()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


t_dense = timed(dense)
full_plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jobs, starts,
                      n, n, nJ, tg)
t_full = timed(lambda: ls_g.enqueue(full_plan))
print(f'one GPU: value + gradient solvers {t_full:.3f} ms, dense part '
      f'{t_dense:.3f} ms, step {t_full + t_dense:.3f} ms')
for world in worlds:
    spv = measured_shard_plan(b, G, kn, ke, jobs, n, n, tv, 0, world)
    spg = measured_shard_plan(b, G, kn, ke, jobs, n, n, tg, 0, world)
    rows = []
    for r in range(world):
        jv = np.ascontiguousarray(jobs[spv.shards[r]])
        jg = np.ascontiguousarray(jobs[spg.shards[r]])
        slab = torch.zeros(len(jg) * (1 + nJ), dtype=torch.float64
                           if real is np.float64 else torch.float32, device=dev)
        pv = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jv, starts,
                       n, n, nJ, tv, packed=True, merge_map=spv.merge_map)
        pg = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, jg, starts,
                       n, n, nJ, tg, packed=True, merge_map=spg.merge_map,
                       gramian_ptr=slab.data_ptr(),
                       gradient_ptr=slab.data_ptr()
                       + len(jg) * slab.element_size())
        ti = torch.as_tensor(jg['i'].astype(np.int64), device=dev)
        tj = torch.as_tensor(jg['j'].astype(np.int64), device=dev)
        mult = torch.where(ti == tj, 1.0, 2.0).to(torch.float64)
        rows_g = slab[len(jg):].view(len(jg), nJ)

        def contraction(W):
            return (rows_g.to(torch.float64)
                    * (W[ti, tj] * mult).unsqueeze(1)).sum(dim=0)

        def serial():
            ls_g.enqueue(pg)                 # (the null stream waits for it)
            return contraction(dense())

        def overlapped():
            ls_v.enqueue(pv)
            ls_g.enqueue(pg, detached=True)
            W = dense()
            ls_g.join()
            return contraction(W)

        rows.append((timed(lambda: ls_g.enqueue(pg)),
                     timed(lambda: ls_v.enqueue(pv)),
                     timed(serial), timed(overlapped)))
    a = np.array(rows)
    ideal = (t_full + t_dense) / world
    print(f'world {world}: per rank (max over ranks) shard {a[:, 0].max():.3f} '
          f'values {a[:, 1].max():.3f} | serial {a[:, 2].max():.3f} ms '
          f'(efficiency {ideal / a[:, 2].max():.2f}) | overlapped '
          f'{a[:, 3].max():.3f} ms (efficiency {ideal / a[:, 3].max():.2f}) '
          f'| ideal {ideal:.3f}')
    print('      serial per rank    ', np.round(a[:, 2], 3))
    print('      overlapped per rank', np.round(a[:, 3], 3))
