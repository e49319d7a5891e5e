#!/usr/bin/env python
"""Gram-matrix throughput of the marginalized graph kernel on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--graphs 1000] [--dtype f64|f32] [--gradient]

A *step* is one full pass of the hot path over the batch: every pair of the
symmetric Gram matrix of the synthetic QM7-like set (config 3 of
BASELINE.json) is solved from device-resident graphs, job list
and hyperparameters into the device-resident result.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) the pairs are sharded over the ranks
and one RCCL all-gather per step reassembles the packed results.

The default arithmetic is fp64, as BASELINE.json names it for this
configuration; the reference's CUDA solver computes in fp32, and a short fp32
measurement of the same step is reported in the same line ("other_arithmetic").

Prints ONE JSON line on rank 0 (contract in the task statement) with the
extra objects "roofline" (dominant kernel, HIP-event timed inside the timed
region) and "cpu_baseline" (the C oracle timed on this host, rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC (RCCL / cross-process tensors)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--graphs', type=int, default=1000)
    ap.add_argument('--dtype', default='f64', choices=['f32', 'f64'],
                    help='arithmetic of the solver: f64 is what BASELINE.json '
                         'names for the QM7-1000 configuration (default); '
                         'f32 is the arithmetic of the reference CUDA solver')
    ap.add_argument('--no-f32', action='store_true',
                    help='skip the short fp32 measurement reported next to '
                         'an fp64 run')
    ap.add_argument('--gradient', action='store_true',
                    help='also evaluate dK/dtheta (config 5 kernel part)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--serial', action='store_true',
                    help='all solver variants on one stream (default: one '
                         'HIP stream per variant so short launches fill the '
                         'tails of long ones)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    return ap.parse_args()


def algorithmic_bytes(arena, ji, jj, rsize, n_cols):
    """compulsory HBM bytes of a pair = both graph images (as
    packed in HBM) + the result."""
    blob = np.diff(np.concatenate((arena.blob_start, [arena.nbytes])))
    hdr = 32
    return (blob[ji] + blob[jj] + 2 * hdr + rsize * n_cols).astype(np.int64)


def algorithmic_flops(n_node, n_nz, ji, jj, iters, Fv=9, Fe=8):
    """SURVEY 8(d) with cached edge/node kernel tables:
    k (2 nnzx + 17 N) + nnzx F_e + N F_v."""
    N = n_node[ji] * n_node[jj]
    nnzx = n_nz[ji] * n_nz[jj]
    return iters * (2 * nnzx + 17 * N) + nnzx * Fe + N * Fv


def measure_other_arithmetic(name, graphs, knode, kedge, q, jobs, starts, n,
                             steps, warmup, device):
    """Short measurement of the same step in the other arithmetic (N = 1):
    reported next to the headline number, never instead of it."""
    from graphdot_amd.hip import runtime
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
    real = np.float32 if name == 'f32' else np.float64
    backend = HIPBackend(device=device, real=real)
    kernel = MarginalizedGraphKernel(knode, kedge, q=q, backend=backend)
    plan = backend.prepare(graphs, knode, kedge, kernel.p, kernel.q,
                           kernel.eps, kernel.ftol, kernel.gtol, jobs, starts,
                           n, n, kernel.n_dims,
                           kernel.traits(symmetric=True))
    for _ in range(max(warmup, 1)):
        backend.launch(plan)
    runtime.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        backend.launch(plan)
        runtime.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {'dtype': name, 'value': len(jobs) / dt, 'unit': 'graph-pairs/s',
            'ms_per_step': 1e3 * dt, 'steps': steps,
            'note': 'same step, same graphs, solver built for the other '
                    'arithmetic (f32 = the reference CUDA solver\'s)'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    import cases
    from graphdot_amd.hip import runtime
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
    from graphdot_amd.kernel.marginalized._sharded import ShardPlan

    dist = torch = None
    host_collective = False
    if world > 1:
        import torch
        import torch.distributed as dist
        n_dev = torch.cuda.device_count()
        if n_dev >= world:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device(
                'cuda', local_rank))
        else:
            # fewer GPUs than ranks (development box): ranks share devices
            # and the all-gather goes through gloo on host memory
            host_collective = True
            local_rank = local_rank % max(n_dev, 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group('gloo')
    real = np.float32 if args.dtype == 'f32' else np.float64
    backend = HIPBackend(device=local_rank, real=real, record_iterations=True)

    graphs = cases.config3_graphs(args.graphs)
    knode, kedge, q = cases.config3_kernels()
    kernel = MarginalizedGraphKernel(knode, kedge, q=q, backend=backend)
    n = len(graphs)
    i, j = np.triu_indices(n)
    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
    all_jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)
    n_pairs = len(all_jobs)
    traits = kernel.traits(symmetric=True, eval_gradient=args.gradient)
    nJ = kernel.n_dims
    n_cols = 1 + (nJ if args.gradient else 0)
    starts = np.arange(n + 1, dtype=np.uint32)

    # ---- shard (world == 1: the single shard is the whole job list) ---------
    dgs = [backend._register_graph(g) for g in graphs]
    n_node = np.array([g.n_node for g in dgs], dtype=np.int64)
    n_nz = np.array([g.n_nz for g in dgs], dtype=np.int64)
    shard = ShardPlan(i, j, n_node, n_nz, n, n, True, rank, world)
    local_jobs = all_jobs[shard.local] if world > 1 else all_jobs
    local_out = gathered = None
    out_ptrs = {}
    if world > 1:
        # every rank's packed slab [cap values | cap*nJ gradient entries] is
        # written by the kernels straight into the all-gather input
        cap = shard.capacity
        tdtype = torch.float32 if real is np.float32 else torch.float64
        local_out = torch.zeros(cap * n_cols, dtype=tdtype, device='cuda')
        gathered = torch.empty(world * cap * n_cols, dtype=tdtype,
                               device='cuda')
        rs_ = np.dtype(real).itemsize
        out_ptrs = dict(gramian_ptr=local_out.data_ptr(),
                        gradient_ptr=local_out.data_ptr() + cap * rs_)
        torch.cuda.synchronize()
    plan = backend.prepare(graphs, knode, kedge, kernel.p, kernel.q,
                           kernel.eps, kernel.ftol, kernel.gtol, local_jobs,
                           starts, n, n, nJ, traits, packed=(world > 1),
                           **out_ptrs)

    # device-side reassembly of the gathered slabs into the F-order matrix
    # (+ mirror; + one plane per gradient column): part of every step, so
    # that a step ends with the same product at any N
    t_src = t_dst = K_dev = None
    if world > 1:
        src_idx, dst_idx = shard.reassembly_index(n_cols - 1)
        t_src = torch.from_numpy(src_idx).cuda()
        t_dst = torch.from_numpy(dst_idx).cuda()
        K_dev = torch.zeros(n_cols * n * n, dtype=tdtype, device='cuda')
        torch.cuda.synchronize()

    kernel_ms = np.zeros(len(plan.launches))
    nL = len(plan.launches)
    streams = [runtime.Stream() for _ in plan.launches] \
        if not args.serial else None
    # one (start, stop) event pair per timed step and launch: durations are
    # read after the timed region, so a step never waits on the host
    ev_sets = [[(runtime.Event(), runtime.Event()) for _ in range(nL)]
               for _ in range(args.steps + 1)]
    ev_join = runtime.Event()    # previous step (and its all-gather) is done

    def step(index, first=False):
        """One pass.  Everything is enqueued without host synchronisation;
        device-side events order the passes: every solver stream waits for
        the join of the previous pass, the join (null stream, which also
        runs the collective and the reassembly) waits for every solver."""
        ev = ev_sets[index]
        if streams is not None:
            for k, L in enumerate(plan.launches):
                if not first:
                    streams[k].wait_event(ev_join)
                ev[k][0].record(streams[k].h)
                runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                               stream=streams[k].h,
                               dynamic_lds=L['dynamic_lds'])
                ev[k][1].record(streams[k].h)
            for k in range(nL):
                runtime.null_stream_wait_event(ev[k][1])
        else:
            for k, L in enumerate(plan.launches):
                ev[k][0].record()
                runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                               dynamic_lds=L['dynamic_lds'])
                ev[k][1].record()
        if world > 1:
            if host_collective:
                h = local_out.cpu()
                g = torch.empty(world * h.numel(), dtype=h.dtype)
                dist.all_gather_into_tensor(g, h)
                gathered.copy_(g)
            else:
                dist.all_gather_into_tensor(gathered, local_out)
            K_dev.index_copy_(0, t_dst, gathered.index_select(0, t_src))
        ev_join.record()

    def sync():
        runtime.synchronize()
        if world > 1:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    for w in range(args.warmup):
        step(args.steps, first=(w == 0))
    if args.warmup == 0:
        ev_join.record()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for it in range(args.steps):
        step(it)
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device='cpu' if host_collective else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # per-kernel durations: HIP events on the stream each kernel ran on
    for it in range(args.steps):
        for k in range(nL):
            kernel_ms[k] += ev_sets[it][k][0].elapsed_ms(ev_sets[it][k][1])
    kernel_ms /= max(args.steps, 1)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    sharded_check = None
    if world > 1:
        # reassemble the gathered slabs and compare a sample with the oracle
        from oracle import mgk
        K = K_dev[:n * n].cpu().numpy().reshape(n, n, order='F')
        K_host = shard.assemble(gathered.cpu().numpy().reshape(
            world, -1)[:, :shard.capacity].ravel())
        assert np.array_equal(K, K_host), 'device and host reassembly differ'
        rng = np.random.default_rng(1)
        probe = rng.choice(n_pairs, size=min(3000, n_pairs), replace=False)
        batch = mgk.TensorProductBatch(graphs, knode, kedge)
        ref, _ = batch.run(i[probe], j[probe], q=q, real='f64', tol=1e-13)
        sharded_check = {
            'max_rel_diff_vs_oracle': float(np.max(np.abs(
                K[i[probe], j[probe]] / ref - 1))),
            'symmetric': bool(np.count_nonzero(K - K.T) == 0),
            'collective': 'gloo/host' if host_collective else 'nccl(RCCL)'}

    ms_per_step = 1e3 * elapsed / args.steps
    value = n_pairs / (elapsed / args.steps)

    # ---- roofline of the dominant kernel -------------------------------------
    arena = plan.keep[0]
    dom = int(np.argmax(kernel_ms))
    L = plan.launches[dom]
    ids = plan.order_host[L['offset']:L['offset'] + L['count']]
    lj = local_jobs[ids]
    lji, ljj = lj['i'].astype(np.int64), lj['j'].astype(np.int64)
    iters = backend.iterations(plan)[ids].astype(np.int64)
    rsize = np.dtype(real).itemsize
    abytes = int(algorithmic_bytes(arena, lji, ljj, rsize, n_cols).sum())
    aflops = int(algorithmic_flops(n_node, n_nz, lji, ljj, iters).sum()
                 * (2 if args.gradient else 1))
    dur = kernel_ms[dom] * 1e-3
    v = L['variant']
    roofline = {
        'bound': 'hbm', 'kernel': backend.kernel_name(v, plan.C, False, L.get('tab', False)),
        'achieved': abytes / dur / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
        'frac': abytes / dur / 1e9 / 8000.0, 'traffic': None,
        'algorithmic_bytes_per_launch': abytes,
        'pairs_per_launch': int(L['count']),
        'avg_launch_ms': float(kernel_ms[dom]),
        'launch_streams': 'one per solver variant (durations overlap)'
                          if not args.serial else 'single stream',
        'note': 'the solver is LDS/VALU-bound by design (CG vectors and the '
                'product-graph operator live in LDS/registers); see compute',
    }
    # measured HBM traffic of that kernel, if a PMC profile of this command
    # has been committed (scripts/profile.sh + scripts/summarize_profile.py)
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            tr = json.load(f)[args.dtype]['kernels'].get(roofline['kernel'])
        if tr and not args.gradient and world == 1:
            roofline['traffic'] = tr['hbm_bytes_per_launch']
            roofline['traffic_source'] = 'profiles/traffic.json (rocprofv3 ' \
                'FETCH_SIZE + WRITE_SIZE, separate passes)'
    except (OSError, KeyError, ValueError):
        pass
    peak_tf = 157.3 if real is np.float32 else 78.6
    compute = {
        'bound': 'valu', 'achieved': aflops / dur / 1e12, 'peak': peak_tf,
        'unit': 'TFLOP/s', 'frac': aflops / dur / 1e12 / peak_tf,
        'algorithmic_flops_per_launch': aflops,
        'mean_cg_iterations': float(iters.mean()),
    }
    # LDS traffic model of SURVEY 8(d): per CG iteration 2 reals per product-
    # graph nonzero (gather + U) and 10 per row, against the aggregate LDS
    # rate of the access width in use (MI355X_MICROARCH.md, LDS: ~75 TB/s
    # for ds_read_b32, ~150 TB/s for ds_read_b64)
    N_ = n_node[lji] * n_node[ljj]
    nnzx_ = n_nz[lji] * n_nz[ljj]
    lds_bytes = int((iters * (2 * nnzx_ + 10 * N_)).sum() * rsize
                    * (2 if args.gradient else 1))
    lds_peak = 75.0 if rsize == 4 else 150.0
    lds = {'bound': 'lds', 'achieved': lds_bytes / dur / 1e12,
           'peak': lds_peak, 'unit': 'TB/s',
           'frac': lds_bytes / dur / 1e12 / lds_peak,
           'algorithmic_lds_bytes_per_launch': lds_bytes,
           'note': 'durations of concurrent launches overlap: the dominant '
                   'kernel shares the chip with the other variants'}
    per_kernel = [
        {'kernel': backend.kernel_name(l['variant'], plan.C, False,
                                       l.get('tab', False)),
         'pairs': int(l['count']), 'grid': int(l['grid']),
         'avg_ms': float(ms)} for l, ms in zip(plan.launches, kernel_ms)]

    # ---- CPU baseline (oracle, 1 core, bounded sample), N = 1 only -----------
    cpu = None
    if world == 1 and not args.no_cpu_baseline and not args.gradient:
        from oracle import mgk
        batch = mgk.TensorProductBatch(graphs, knode, kedge)
        rng = np.random.default_rng(0)
        probe = rng.choice(n_pairs, size=min(2000, n_pairs), replace=False)
        t1 = time.perf_counter()
        batch.run(i[probe], j[probe], q=q, real=args.dtype)
        rate = len(probe) / (time.perf_counter() - t1)
        size = int(min(n_pairs, max(2000, rate * args.cpu_seconds)))
        sample = rng.choice(n_pairs, size=size, replace=False)
        t1 = time.perf_counter()
        ref, _ = batch.run(i[sample], j[sample], q=q, real=args.dtype)
        dt = time.perf_counter() - t1
        cpu = {'value': size / dt, 'unit': 'graph-pairs/s', 'cores': 1,
               'kind': 'port',
               'sample': f'{size} uniformly sampled pairs of the same '
                         f'{n}-graph set, oracle/mgk_oracle.c '
                         f'mgk_gram_tp_{args.dtype}, 1 thread'}
        # same restatement, OpenMP over the pairs, every core of this host
        # 
        try:
            ncore = len(os.sched_getaffinity(0))
            size_all = int(min(n_pairs, max(4000, 0.5 * ncore * rate
                                            * args.cpu_seconds / 2)))
            sample_all = rng.choice(n_pairs, size=size_all, replace=False)
            batch.run(i[probe], j[probe], q=q, real=args.dtype, omp=True)
            t1 = time.perf_counter()
            batch.run(i[sample_all], j[sample_all], q=q, real=args.dtype,
                      omp=True)
            cpu['all_cores'] = {
                'value': size_all / (time.perf_counter() - t1),
                'unit': 'graph-pairs/s', 'cores': ncore,
                'sample': f'{size_all} pairs, OpenMP dynamic schedule'}
        except Exception as e:                      # no libgomp etc.
            cpu['all_cores'] = {'error': str(e)}
        # the reference's own Python CPU path cannot travel to this host; its
        # rate was recorded in the build container (provenance in the file)
        try:
            with open(os.path.join(ROOT, 'profiles',
                                   'r01_reference_python_cpu.json')) as f:
                rp = json.load(f)
            cpu['reference_python'] = {
                k: rp[k] for k in ('value', 'unit', 'cores', 'what', 'where')}
        except (OSError, KeyError, ValueError):
            pass
        # the sample doubles as an on-line parity check of the timed result
        got, _ = backend.collect(plan)
        K = got.reshape(n, n, order='F')
        err = np.max(np.abs(K[i[sample], j[sample]] / ref - 1))
        cpu['max_rel_diff_vs_gpu'] = float(err)

    other = None
    if world == 1 and args.dtype == 'f64' and not args.no_f32 \
            and not args.gradient:
        other = measure_other_arithmetic(
            'f32', graphs, knode, kedge, q, all_jobs, starts, n,
            max(args.steps // 2, 3), args.warmup, local_rank)

    line = {
        'metric': 'graph-pairs/sec (Gram matrix)', 'value': value,
        'unit': 'graph-pairs/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {
            'workload': f'QM7-like synthetic set ({n} molecules, seed 7165, '
                        f'{n_pairs} pairs incl. diagonal), TensorProduct '
                        'atom/bond microkernels, q=0.01, '
                        + ('fp64' if args.dtype == 'f64' else 'fp32')
                        + (', value + gradient' if args.gradient else ''),
            'graphs': n, 'pairs': n_pairs,
            'parallelism': f'pair-sharded x{world}' if world > 1 else 'single',
        },
        'roofline': roofline, 'compute': compute, 'lds': lds,
        'kernels': per_kernel,
        'cpu_baseline': cpu, 'sharded_check': sharded_check,
        'other_arithmetic': other,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
