"""Pair-sharded Gram matrix over the GPUs of one node.

New capability (the reference has no multi-GPU code):
every graph pair is independent (reference job loop: template.cu:57-70), so the
job list is dealt to the ranks in cost order, every rank solves its shard into
a *packed* per-job slab (the output shape of the reference's
``alt_graph_kernel_solver``, experimental/alterantive_mgk/template.cu:71,103-118)
and one all-gather of equal-sized slabs reassembles the matrix.  There is no
other data-path collective.

The class is transport- and device-agnostic: ``solve_local(job_ids)`` computes
the shard (the HIP plan on a GPU, the oracle in the CPU tests) and
``all_gather(local)`` is ``torch.distributed.all_gather_into_tensor`` over RCCL
(backend "nccl") on GPUs or gloo on CPU.
"""
import numpy as np


def predict_cost(n_node, n_nz, ji, jj):
    """Relative cost of a pair: product-graph nonzeros plus vector work."""
    return n_nz[ji] * n_nz[jj] + 4 * n_node[ji] * n_node[jj]


def partition(cost, world_size):
    """Deal jobs to ranks in descending-cost snake order (LPT-like).  Returns
    a list of job-id arrays, one per rank, each sorted by descending cost."""
    order = np.argsort(-cost, kind='stable')
    k = np.arange(len(order))
    rnd, pos = k // world_size, k % world_size
    rank = np.where(rnd % 2 == 0, pos, world_size - 1 - pos)
    return [order[rank == r] for r in range(world_size)]


class ShardPlan:
    """Static description of one rank's share and of the reassembly."""

    def __init__(self, ji, jj, n_node, n_nz, nX, nY, symmetric, rank,
                 world_size):
        self.rank, self.world_size = rank, world_size
        self.ji, self.jj = np.asarray(ji), np.asarray(jj)
        self.nX, self.nY, self.symmetric = nX, nY, symmetric
        cost = predict_cost(np.asarray(n_node), np.asarray(n_nz),
                            self.ji, self.jj)
        self.shards = partition(cost, world_size)
        self.capacity = max(len(s) for s in self.shards) if len(cost) else 0
        self.local = self.shards[rank]
        # position of every job in the gathered [world_size, capacity] slab
        self.slot = np.empty(len(cost), dtype=np.int64)
        for r, s in enumerate(self.shards):
            self.slot[s] = r * self.capacity + np.arange(len(s))

    def scatter_index(self, starts_x, starts_y):
        """Flat F-order destinations (and mirrored destinations) of the
        gathered slab entries -- K(I1, I2) = gathered[slot]
        (tensor_view column-major: graphdot/cpp/tensor_view.h:22-33)."""
        a = starts_x[self.ji].astype(np.int64)
        b = starts_y[self.jj].astype(np.int64)
        dst = a + self.nX * b
        if self.symmetric:
            off = self.ji != self.jj
            mdst = b[off] + self.nX * a[off]
            return dst, self.slot, mdst, self.slot[off]
        return dst, self.slot, None, None

    def reassembly_index(self, n_grad=0):
        """(src, dst) flat indices that turn the all-gathered slabs into the
        column-major result on the device with one gather + one scatter
        (``out[dst] = gathered[src]``).  Every rank's slab is laid out as
        ``[capacity values | capacity * n_grad gradient entries, job-major]``
        (what the packed solver writes: mgk_solver.h, F_PACKED); the result is
        ``[nX * nY values | n_grad planes of nX * nY]`` with the mirrored
        entries of a symmetric matrix filled in."""
        n_cols = 1 + n_grad
        sx = np.arange(self.nX + 1, dtype=np.int64)
        jy = self.jj if self.symmetric else self.jj - self.nX
        saved, self.jj = self.jj, jy
        try:
            dst, slot, mdst, mslot = self.scatter_index(sx, sx if self.symmetric
                                                        else np.arange(self.nY + 1))
        finally:
            self.jj = saved
        if mdst is not None:
            dst = np.concatenate((dst, mdst))
            slot = np.concatenate((slot, mslot))
        cap = self.capacity
        rank, pos = slot // cap, slot % cap
        base = rank * cap * n_cols
        srcs, dsts = [base + pos], [dst]
        for c in range(n_grad):
            srcs.append(base + cap + pos * n_grad + c)
            dsts.append((c + 1) * self.nX * self.nY + dst)
        return np.concatenate(srcs), np.concatenate(dsts)

    def assemble(self, gathered, n_cols=1):
        """Host reassembly of the gathered slabs into the (nX, nY[, n_cols])
        matrix (graph-level outputs: starts are 0..n)."""
        sx = np.arange(self.nX + 1)
        sy = np.arange(self.nY + 1) if not self.symmetric else sx
        jy = self.jj if self.symmetric else self.jj - self.nX
        saved = self.jj
        self.jj = jy
        dst, src, mdst, msrc = self.scatter_index(sx, sy)
        self.jj = saved
        g = np.asarray(gathered).reshape(self.world_size * self.capacity,
                                         n_cols)
        out = np.zeros((self.nX * self.nY, n_cols), dtype=g.dtype)
        out[dst] = g[src]
        if mdst is not None:
            out[mdst] = g[msrc]
        out = out.reshape(self.nY, self.nX, n_cols).transpose(1, 0, 2)
        return out[:, :, 0] if n_cols == 1 else out


def distributed_backend(**kwargs):
    """A HIPBackend that shards every graph-level evaluation over the ranks
    of the initialised ``torch.distributed`` process group (one process per
    GPU, launched e.g. by ``python -m torch.distributed.run``):

        kernel = MarginalizedGraphKernel(knode, kedge, q=q,
                                         backend=distributed_backend())
        K = kernel(graphs)          # every rank gets the full matrix

    Pairs are independent, so each rank solves its cost-balanced share of the
    job list into a packed slab (`ShardPlan`), one all-gather of equal-sized
    slabs follows (RCCL for the "nccl" backend, host memory for "gloo"), and
    every rank scatters the gathered values into the caller's arrays.  Nodal
    and diagonal evaluations, and runs without a process group, take the
    single-GPU path.  Keyword arguments as for HIPBackend."""
    from ._backend_hip import HIPBackend

    class DistributedHIPBackend(HIPBackend):

        def __init__(self, **kw):
            super().__init__(**kw)
            self._shard_plans = {}

        @staticmethod
        def shards_over_ranks():
            """True when a process group with more than one rank is up."""
            import torch.distributed as dist
            return (dist.is_available() and dist.is_initialized()
                    and dist.get_world_size() > 1)

        def _shard_plan(self, dgraphs, jobs, nX, nY, symmetric, rank, world):
            key = (tuple(map(id, dgraphs)), id(jobs) if not
                   jobs.flags.writeable else hash(jobs.tobytes()),
                   nX, nY, symmetric, rank, world)
            hit = self._shard_plans.get(key)
            if hit is None:
                if len(self._shard_plans) > 8:
                    self._shard_plans.clear()
                n_node = np.array([g.n_node for g in dgraphs], np.int64)
                n_nz = np.array([g.n_nz for g in dgraphs], np.int64)
                sp = ShardPlan(jobs['i'].astype(np.int64),
                               jobs['j'].astype(np.int64), n_node, n_nz,
                               nX, nY, symmetric, rank, world)
                hit = self._shard_plans[key] = (sp, jobs, list(dgraphs), {})
            return hit

        def __call__(self, graphs, node_kernel, edge_kernel, p, q, eps, ftol,
                     gtol, jobs, starts, gramian, gradient, nX, nY, nJ,
                     traits, timer):
            import torch
            import torch.distributed as dist
            graph_level = (traits.nodal is False and not traits.diagonal)
            if not (self.shards_over_ranks() and graph_level):
                return super().__call__(
                    graphs, node_kernel, edge_kernel, p, q, eps, ftol, gtol,
                    jobs, starts, gramian, gradient, nX, nY, nJ, traits,
                    timer)
            rank, world = dist.get_rank(), dist.get_world_size()
            jobs = np.ascontiguousarray(jobs)
            dgraphs = [self._register_graph(g) for g in graphs]
            sp, _, _, index_cache = self._shard_plan(
                dgraphs, jobs, int(nX), int(nY), bool(traits.symmetric), rank,
                world)
            n_grad = int(nJ) if traits.eval_gradient is True else 0
            n_cols = 1 + n_grad
            timer.tic('GPU kernel execution')
            local_jobs = jobs[sp.local]
            local_jobs.flags.writeable = True      # fresh array: by checksum
            plan = self.prepare(graphs, node_kernel, edge_kernel, p, q, eps,
                                ftol, gtol, local_jobs, starts, nX, nY, nJ,
                                traits, timer, packed=True)
            self.launch(plan)
            values, grads = self.collect(plan)
            timer.toc('GPU kernel execution')

            # slab of this rank: [capacity values | capacity * n_grad entries]
            cap = sp.capacity
            slab = np.zeros(cap * n_cols, dtype=values.dtype)
            slab[:len(values)] = values
            if n_grad:
                slab[cap:cap + len(grads)] = grads
            on_device = dist.get_backend() == 'nccl'
            t_slab = torch.from_numpy(slab)
            if on_device:
                t_slab = t_slab.cuda()
            gathered = torch.empty(world * len(slab), dtype=t_slab.dtype,
                                   device=t_slab.device)
            dist.all_gather_into_tensor(gathered, t_slab)
            gathered = gathered.cpu().numpy()

            if n_grad not in index_cache:
                index_cache[n_grad] = sp.reassembly_index(n_grad)
            src, dst = index_cache[n_grad]
            n_out = int(nX) * int(nY)
            gramian[:] = 0
            gramian[dst[dst < n_out]] = gathered[src[dst < n_out]]
            if n_grad:
                sel = dst >= n_out
                gradient[:] = 0
                gradient[dst[sel] - n_out] = gathered[src[sel]]

    return DistributedHIPBackend(**kwargs)
