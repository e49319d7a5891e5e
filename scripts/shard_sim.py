#!/usr/bin/env python3
"""What one rank of an N-GPU run computes, timed on one GPU: for every rank of
a world of 2 / 4 / 8 the local shard of the 1000-graph matrix is prepared and
its step timed; max over ranks against (full step) / N is the compute part of
the strong-scaling efficiency (the all-gather is not in it).
    python scripts/shard_sim.py [--f32] [--mode=snake|blocks|measured] [--gradient] [--no-merge]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend, LaunchSet
from graphdot_amd.kernel.marginalized._sharded import (
    ShardPlan, partition, predict_cost, measured_shard_plan)

real = np.float32 if '--f32' in sys.argv else np.float64
grad = '--gradient' in sys.argv
modes = [a.split('=')[1] for a in sys.argv if a.startswith('--mode=')] or ['snake', 'measured']
n = 1000
G = cases.config3_graphs(n)
kn, ke, q = cases.config3_kernels()
b = HIPBackend(real=real, min_launch=0 if '--no-merge' in sys.argv else 8192)
k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
i, j = np.triu_indices(n)
jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
starts = np.arange(n + 1, dtype=np.uint32)
traits = k.traits(symmetric=True, eval_gradient=grad)
dgs = [b._register_graph(g) for g in G]
n_node = np.array([g.n_node for g in dgs], np.int64)
n_nz = np.array([g.n_nz for g in dgs], np.int64)
cost = predict_cost(n_node, n_nz, i.astype(np.int64), j.astype(np.int64))
ls = LaunchSet()


def step_ms(local_jobs, steps=30, merge_map=None):
    plan = b.prepare(G, kn, ke, k.p, k.q, k.eps, k.ftol, k.gtol, local_jobs,
                     starts, n, n, k.n_dims, traits, packed=True,
                     merge_map=merge_map)
    for _ in range(3):
        ls.enqueue(plan)
    runtime.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ls.enqueue(plan)
    runtime.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, len(plan.launches)


full, nl = step_ms(jobs)
print(f'full step {full:.3f} ms, {nl} launches')
for world in (2, 4, 8):
    for mode in modes:
        mm = None
        if mode == 'measured':
            # contiguous blocks of the launch order with equal measured time,
            # launch merging decided on the whole list
            # (_sharded.measured_shard_plan: what ShardedStep uses)
            sp = measured_shard_plan(b, G, kn, ke, jobs, n, n, traits, 0,
                                     world)
            shards, mm = sp.shards, sp.merge_map
        else:
            shards = partition(cost, world, mode)
        t = [step_ms(np.ascontiguousarray(jobs[s]), merge_map=mm)
             for s in shards]
        ms = np.array([x[0] for x in t])
        if mode == 'measured':
            print('      predicted  ', np.round(np.array(sp.predicted) * 1e-6, 3))
            # what DistributedHIPBackend does when it builds a step: the ranks
            # time their shards, all-gather the times, take the cuts again
            for rnd in range(2):
                sp = sp.rebalanced(ms)
                shards = sp.shards
                t = [step_ms(np.ascontiguousarray(jobs[s]), merge_map=mm)
                     for s in shards]
                ms = np.array([x[0] for x in t])
                print(f'      rebalance round {rnd + 1}: max {ms.max():.3f} '
                      f'mean {ms.mean():.3f} efficiency '
                      f'{full / world / ms.max():.2f}', np.round(ms, 3))
        print(f'world {world} {mode:6s}: max {ms.max():.3f} mean {ms.mean():.3f} '
              f'ideal {full / world:.3f}  efficiency {full / world / ms.max():.2f}  '
              f'launches {[x[1] for x in t]}  pairs {[len(s) for s in shards]}')
        print('      per rank ms', np.round(ms, 3))
