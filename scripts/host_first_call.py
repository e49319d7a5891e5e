#!/usr/bin/env python3
"""Host side of a first call, no device: packing, arena, classification and
job layout of the 1000-graph benchmark set, timed piece by piece."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import cases
from graphdot_amd.hip import hostlib
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
real = np.float64 if '--f64' in sys.argv else np.float32
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()
hostlib.lib()
for trial in range(3):
    for g in G:
        for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
            del g.cookie[key]
    b = HIPBackend(real=real)
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    t0 = time.perf_counter()
    jobs = hostlib.pairwise_jobs(1000, None, np.dtype([('i', np.uint32), ('j', np.uint32)]))
    t1 = time.perf_counter()
    dgraphs, ek, C, fields = b._graphs_and_kernels(G, kn, ke, k.traits(symmetric=True))
    t2 = time.perf_counter()
    arena = b._host_arena(dgraphs, fields)
    t3 = time.perf_counter()
    out = b._partition(dgraphs, jobs, C, b._table_bytes(arena), b._global_tables(arena))
    t4 = time.perf_counter()
    img = arena.relocated(1 << 40)
    t5 = time.perf_counter()
    print(f'trial {trial}: jobs {1e3*(t1-t0):.2f}  pack {1e3*(t2-t1):.2f}  arena {1e3*(t3-t2):.2f}  '
          f'partition {1e3*(t4-t3):.2f}  relocate {1e3*(t5-t4):.2f}  total {1e3*(t5-t0):.2f} ms')
if '--profile' in sys.argv:
    import cProfile, pstats
    for g in G:
        for key in [k_ for k_ in g.cookie if k_ != 'rowtypes']:
            del g.cookie[key]
    b = HIPBackend(real=real)
    pr = cProfile.Profile(); pr.enable()
    dgraphs, ek, C, fields = b._graphs_and_kernels(G, kn, ke, k.traits(symmetric=True))
    arena = b._host_arena(dgraphs, fields)
    out = b._partition(dgraphs, jobs, C, b._table_bytes(arena), b._global_tables(arena))
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(25)
