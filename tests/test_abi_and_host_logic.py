"""CPU-only checks of the drop-in boundary and of the backend's host logic:
the C-ABI library builds, loads and exports every symbol of include/gdhip.h;
code generation, job partitioning and the pair sharding are exercised without
a device (no compute calls)."""
import ctypes
import os
import re
import sys
import numpy as np
import pytest
import cases
from graphdot_amd.hip import jit, runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import (
    HIPBackend, VARIANTS, Variant, declstruct)
from graphdot_amd.kernel.marginalized._sharded import ShardPlan, partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'gdhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gd_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    L = runtime.lib()
    names = declared_symbols()
    assert len(names) >= 28
    for name in names:
        assert hasattr(L, name), name
        assert name in runtime.SIGNATURES or name in (
            'gd_last_error', 'gd_version'), f'{name} has no ctypes signature'
    assert b'gfx950' in L.gd_version()


def test_host_library_exports_every_declared_symbol():
    """include/gdhost.h (packer, label classes, classification, job layout)
    against libgdhost.so and its ctypes signatures; bad arguments are
    reported, not crashed on."""
    from graphdot_amd.hip import hostlib
    text = open(os.path.join(ROOT, 'include', 'gdhost.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = sorted(set(re.findall(r'\b(gdh_[a-z0-9_]+)\s*\(', text)))
    assert len(names) == 9
    L = hostlib.lib()
    for name in names:
        assert hasattr(L, name), name
        assert name in hostlib.SIGNATURES or name == 'gdh_version', name
    assert b'gdhost' in L.gdh_version()
    jobs = np.array([(0, 5)], dtype=[('i', np.uint32), ('j', np.uint32)])
    with pytest.raises(hostlib.HostLibError):      # graph 5 of 2
        hostlib.pair_keys(jobs, np.zeros(2, np.int32), 1)
    with pytest.raises(hostlib.HostLibError):      # rank out of range
        hostlib.order_jobs(np.zeros(3, np.int32), np.array([4], np.int32), 2)
    # the job records travel in launch order
    jobs = np.array([(0, 1), (2, 3), (4, 5), (6, 7)],
                    dtype=[('i', np.uint32), ('j', np.uint32)])
    order, moved = hostlib.order_jobs(np.array([1, 0, 1, 0], np.int32),
                                      np.array([1, 0], np.int32), 2, jobs)
    assert order.tolist() == [0, 2, 1, 3]
    assert np.array_equal(moved, jobs[order])


def test_device_calls_fail_loudly_without_a_gpu():
    """No silent CPU fallback: with no device the ABI reports an error."""
    L = runtime.lib()
    n = ctypes.c_int(-1)
    rc = L.gd_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip('a GPU is present')
    assert rc != 0 or n.value == 0
    with pytest.raises(runtime.HIPError):
        runtime.check(L.gd_init(0))
    assert L.gd_last_error()


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'graphdot_amd')):
        for f in files:
            if f.endswith(('.py', '.h', '.cpp', '.hip')):
                text = open(os.path.join(dirpath, f), errors='replace').read()
                if re.search(r'^\s*(from|import)\s+oracle\b', text, re.M) \
                        or 'mgk_oracle' in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_jit_compiles_for_gfx950_and_caches(tmp_path, monkeypatch):
    monkeypatch.setattr(jit, 'CACHE_DIR', str(tmp_path))
    src = ('#include <hip/hip_runtime.h>\n'
           'extern "C" __global__ void k(float *x) { x[threadIdx.x] = 1.f; }\n')
    path = jit.compile_source(src)
    assert os.path.getsize(path) > 0
    with open(path, 'rb') as f:          # clang offload bundle of one target
        assert b'gfx950' in f.read()
    mtime = os.path.getmtime(path)
    assert jit.compile_source(src) == path
    assert os.path.getmtime(path) == mtime
    with pytest.raises(jit.CompileError):
        jit.compile_source('this is not HIP')


def test_solver_translation_unit_compiles_all_modes():
    """The generated TU (value and value+gradient, float and double) builds
    for gfx950; static_asserts inside tie the host packing to the device
    structs."""
    G = cases.nlw_example_graphs()
    knode, kedge, q = cases.config2a_kernels()
    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
    jobs = np.array([(0, 0), (0, 1), (1, 2)], dtype=job_t)
    for real in (np.float32, np.float64):
        backend = HIPBackend(real=real)
        k = MarginalizedGraphKernel(knode, kedge, q=q, backend=backend)
        for eg in (False, True):
            paths = backend.precompile(
                G, knode, kedge, k.p, jobs,
                k.traits(symmetric=True, eval_gradient=eg))
            assert all(os.path.getsize(p) > 0 for p in paths)


def test_declstruct_names_nested_and_empty_members():
    dt = np.dtype([('weight', np.dtype([])),
                   ('label', np.dtype([('h', np.float32)], align=True))],
                  align=True)
    text = declstruct(dt, 'T')
    assert 'constexpr static _empty weight' in text
    assert 'struct T_label {float32 h;} label;' in text


def test_job_classification_respects_variant_capacity():
    G = cases.config2_graphs(12, seed=1)
    backend = HIPBackend()
    dgs = [backend._register_graph(g) for g in G]
    i, j = np.triu_indices(len(G))
    choice, cost, ntask, gbytes, NP, _ = backend.classify(i, j, dgs, 1)
    n = np.array([d.n_node for d in dgs])
    nz = np.array([d.n_nz for d in dgs])
    assert np.all(choice >= 0)
    for k, (a, b) in enumerate(zip(i, j)):
        v = backend.variants[choice[k]]
        # rows and stage-1 tasks live in index spaces with odd strides
        # (ldp = n2 | 1, ldu = nnz1 + 1: LDS bank spreading, mgk_solver.h)
        assert n[a] * (n[b] | 1) <= 64 * v.W * v.R
        ldu = nz[a] + 1
        assert ntask[k] == max(ldu * n[b], n[a] + n[b] + 2)
        if getattr(v, 'S', 1) == 0:
            continue        # on-the-fly variant: no register slots to fit
        # brute-force the stage-1 walk of mgk_solver.h for this job
        T = 64 * v.W
        deg = dgs[b].adjacency_count
        worst = 0
        for w in range(v.W):
            total, kb = 0, 0
            while kb * T + 64 * w < ldu * n[b]:
                total += max(1, deg[(kb * T + 64 * w) // ldu])
                kb += 1
            worst = max(worst, total)
        assert worst <= v.S
        assert backend.lds_bytes(v, 1, ntask[k], gbytes[k]) <= 160 * 1024
    # with only the smallest variant available, large pairs must be refused
    small = HIPBackend(variants=[Variant(1, 8, 2)])
    with pytest.raises(NotImplementedError):
        small.classify(i, j, dgs, 1)


def test_owner_computes_classification():
    """Molecular graphs (every degree <= 4) go to the owner-computes variants
    (mgk_oc.h); the slot count the host assigns by is the walk of the device
    code: rows sorted by descending degree product, dealt in batches of 64,
    a batch walks the product of its first row."""
    from graphdot_amd.kernel.marginalized._backend_hip import OCVariant
    G = cases.config3_graphs(40, seed=5)
    backend = HIPBackend()
    dgs = [backend._register_graph(g) for g in G]
    i, j = np.triu_indices(len(G))
    choice, cost, ntask, gbytes, NP, gb_oc = backend.classify(i, j, dgs, 1)
    assert np.all(choice >= 0)
    static_seen = set()
    for k, (a, b) in enumerate(zip(i, j)):
        v = backend.variants[choice[k]]
        assert isinstance(v, OCVariant) and v.W == 1
        d1, d2 = dgs[a].adjacency_count, dgs[b].adjacency_count
        assert max(d1.max(), d2.max()) <= v.D
        # brute force: stable sort of the rows by descending product with
        # ties in the rectangle order (d1, d2) ascending, then row-major
        rows = sorted(((-int(x) * int(y), int(x), int(y))
                       for x in d1 for y in d2))
        trips = [-rows[t][0] for t in range(0, len(rows), 64)]     # (W = 1)
        assert sum(trips) <= v.S and len(rows) <= 64 * v.R
        if v.L:
            # static layout (seg_layout in mgk_oc.h): batch k owns exactly
            # L[k] slots, so the first -- the heaviest -- row of every batch
            # must fit its segment
            static_seen.add(v.L)
            assert len(trips) <= len(v.L)
            assert all(t <= cap for t, cap in zip(trips, v.L))
            assert sum(v.L) == v.S and len(v.L) == v.R
        assert backend.lds_bytes(v, 1, NP[k], gb_oc[k]) <= 160 * 1024
    assert len(static_seen) >= 3          # the molecular profiles are static
    # without the static layouts every pair still finds a dynamic variant
    dyn = HIPBackend(variants=[v for v in backend.variants
                               if not (isinstance(v, OCVariant) and v.L)])
    cd, *_ = dyn.classify(i, j, dgs, 1)
    assert all(isinstance(dyn.variants[c], OCVariant)
               and dyn.variants[c].L is None for c in cd)
    # graphs with nodes of degree 5..8 take the D = 8 kernels, and the
    # two-stage solver when the owner-computes menu is switched off
    G2 = cases.config2_graphs(4, nmin=8, nmax=24, seed=1)
    dg2 = [backend._register_graph(g) for g in G2]
    assert max(int(d.adjacency_count.max()) for d in dg2) > 4
    md = [int(d.adjacency_count.max()) for d in dg2]
    c2, *_ = backend.classify(np.array([0, 1]), np.array([2, 3]), dg2, 1)
    for c, (a, b) in zip(c2, ((0, 2), (1, 3))):
        v = backend.variants[c]
        assert isinstance(v, OCVariant)
        assert v.D == (8 if max(md[a], md[b]) > 4 else 4)
    # multi-wave pairs: the 64-row chunks of a batch go to the waves in snake
    # order; a wave walks, per batch, the product of its chunk's first row
    G3 = cases.config2_graphs(10, nmin=24, nmax=48, seed=2)
    dg3 = [backend._register_graph(g) for g in G3]
    i3, j3 = np.triu_indices(len(G3))
    c3, *_ = backend.classify(i3, j3, dg3, 1)
    seen_w = set()
    for c, a, b in zip(c3, i3, j3):
        v = backend.variants[c]
        if not isinstance(v, OCVariant):
            continue
        seen_w.add(v.W)
        prod = np.sort(np.outer(dg3[a].adjacency_count,
                                dg3[b].adjacency_count).ravel())[::-1]
        T, worst = 64 * v.W, 0
        for w in range(v.W):
            total, k = 0, 0
            while True:
                first = k * T + 64 * ((v.W - 1 - w) if k % 2 else w)
                if k * T >= len(prod):
                    break
                if first < len(prod):
                    total += int(prod[first])
                k += 1
            worst = max(worst, total)
        assert worst <= v.S and len(prod) <= T * v.R
    assert seen_w & {4, 8, 16}
    two_stage = HIPBackend(variants=VARIANTS)
    c3, *_ = two_stage.classify(np.array([0, 1]), np.array([2, 3]), dg2, 1)
    assert all(not isinstance(two_stage.variants[c], OCVariant) for c in c3)


def test_partition_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    cost = rng.integers(1, 1000, size=1001)
    shards = partition(cost, 8)
    allj = np.sort(np.concatenate(shards))
    assert allj.tolist() == list(range(1001))
    loads = np.array([cost[s].sum() for s in shards])
    assert loads.max() / loads.mean() < 1.02
    assert max(map(len, shards)) - min(map(len, shards)) <= 1
    # 'blocks': contiguous runs of the cost-sorted list with equal total cost
    blocks = partition(cost, 8, mode='blocks')
    assert np.sort(np.concatenate(blocks)).tolist() == list(range(1001))
    loads = np.array([cost[s].sum() for s in blocks])
    assert loads.max() / loads.mean() < 1.05
    for a, b in zip(blocks, blocks[1:]):     # rank r: costlier pairs than r + 1
        assert cost[a].min() >= cost[b].max()
    assert len(blocks[0]) < len(blocks[-1])


def _shard_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import mgk
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    G = cases.config3_graphs(10, seed=3)
    knode, kedge, q = cases.config3_kernels()
    n = len(G)
    i, j = np.triu_indices(n)
    n_node = np.array([len(g.nodes) for g in G])
    n_nz = np.array([mgk._side(g).nnz for g in G])
    plan = ShardPlan(i, j, n_node, n_nz, n, n, True, rank, world)
    # local shard through the oracle (the GPU path plugs its HIP plan here)
    batch = mgk.TensorProductBatch(G, knode, kedge)
    local, _ = batch.run(i[plan.local], j[plan.local], q=q, real='f64',
                         tol=1e-13)
    slab = torch.zeros(plan.capacity, dtype=torch.float64)
    slab[:len(local)] = torch.from_numpy(local)
    gathered = torch.empty(world * plan.capacity, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, slab)
    K = plan.assemble(gathered.numpy())
    np.save(os.path.join(tmp, f'K{rank}.npy'), K)
    dist.destroy_process_group()


def _balance_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import mgk
    from graphdot_amd.kernel.marginalized._sharded import (
        measured_shard_plan, balance_by_measurement)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    G = cases.config3_graphs(300, seed=3)   # (enough pairs that the
    # launch tails do not pin the cut to a launch boundary)
    knode, kedge, q = cases.config3_kernels()
    b = HIPBackend(real=np.float64)                 # host side only
    k = MarginalizedGraphKernel(knode, kedge, q=q, backend=b)
    jobs = k._pairwise_jobs(len(G))
    plan = measured_shard_plan(b, G, knode, kedge, jobs, len(G), len(G),
                               k.traits(symmetric=True), rank, world)
    n_node = np.array([len(g.nodes) for g in G])
    ji, jj = jobs['i'].astype(int), jobs['j'].astype(int)
    # what this "GPU" really takes per pair: the table is 60 % low for the
    # large pairs
    N = n_node[ji] * n_node[jj]
    real = plan._model['times'] * np.where(N > np.median(N), 1.6, 1.0)

    class FakeStep:
        device = 'cpu'

        def __init__(self, p):
            self.p = p

        def time_local(self):
            return cost(self.p, self.p.rank) * 1e-6

    def cost(p, r):
        # the pairs at their real rate + the launch tails the plan counts
        s = p.shards[r]
        tails = p.predicted[r] - float(p._model['times'][s].sum())
        return float(real[s].sum()) + tails

    before = [cost(plan, r) for r in range(world)]
    step, plan = balance_by_measurement(FakeStep(plan), plan, FakeStep, 3)
    after = [cost(plan, r) for r in range(world)]
    # the re-balanced plan drives the sharded evaluation (oracle as solver)
    batch = mgk.TensorProductBatch(G, knode, kedge)
    local, _ = batch.run(ji[plan.local], jj[plan.local], q=q, real='f64',
                         tol=1e-13)
    slab = torch.zeros(plan.capacity, dtype=torch.float64)
    slab[:len(local)] = torch.from_numpy(local)
    gathered = torch.empty(world * plan.capacity, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, slab)
    K = plan.assemble(gathered.numpy())
    np.savez(os.path.join(tmp, f'bal{rank}.npz'), K=K, before=before,
             after=after, s0=plan.shards[0], s1=plan.shards[1])
    dist.destroy_process_group()


def test_measured_rebalancing_world_size_2_gloo(tmp_path):
    """`balance_by_measurement` with two gloo ranks on CPU: every rank times
    its own shard (here: a stand-in whose large pairs are 60 % slower than the
    cost table says), the times are all-gathered, both ranks take the same new
    cuts, the imbalance goes from > 8 % to < 4 %, and the matrix assembled
    from the re-balanced shards is still the oracle's."""
    import torch.multiprocessing as mp
    from oracle import mgk
    port = 29500 + (os.getpid() + 37) % 1000
    mp.spawn(_balance_worker, args=(2, port, str(tmp_path)), nprocs=2,
             join=True)
    r0, r1 = (np.load(os.path.join(str(tmp_path), f'bal{r}.npz'))
              for r in range(2))
    assert np.array_equal(r0['s0'], r1['s0'])
    assert np.array_equal(r0['s1'], r1['s1'])
    assert sorted(np.concatenate((r0['s0'], r0['s1'])).tolist()) == \
        list(range(300 * 301 // 2))
    imb = lambda t: max(t) / (sum(t) / len(t))        # noqa: E731
    assert imb(r0['before']) > 1.08 and imb(r0['after']) < 1.04, (
        r0['before'], r0['after'])
    G = cases.config3_graphs(300, seed=3)
    knode, kedge, q = cases.config3_kernels()
    probe = np.random.default_rng(0).integers(0, 300, size=(60, 2))
    batch = mgk.TensorProductBatch(G, knode, kedge)
    ref, _ = batch.run(probe[:, 0], probe[:, 1], q=q, real='f64', tol=1e-13)
    for r in (r0, r1):
        K = r['K']
        assert np.allclose(K[probe[:, 0], probe[:, 1]], ref, rtol=1e-9)
        assert np.count_nonzero(K - K.T) == 0


def test_pair_sharding_world_size_2_gloo(tmp_path):
    """The multi-GPU path (shard -> packed slab -> all-gather -> reassembly)
    with two gloo ranks on CPU, local shards computed by the oracle."""
    import torch.multiprocessing as mp
    from oracle import mgk
    port = 29500 + os.getpid() % 1000
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path)), nprocs=2,
             join=True)
    G = cases.config3_graphs(10, seed=3)
    knode, kedge, q = cases.config3_kernels()
    ref = mgk.gram(G, knode, kedge, q=q)
    for r in range(2):
        K = np.load(os.path.join(str(tmp_path), f'K{r}.npy'))
        assert np.allclose(K, ref, rtol=1e-9)
        assert np.count_nonzero(K - K.T) == 0


def test_label_classes_are_numbered_over_the_attributes_the_kernels_read():
    from graphdot_amd.kernel.marginalized._devicegraph import (
        DeviceGraph, GraphArena, class_bytes)
    G = cases.config3_graphs(40, seed=1)
    dgs = [DeviceGraph(g, np.float32) for g in G]
    full = GraphArena(dgs)
    used = GraphArena(dgs, ('aromatic', 'atomic_number', 'hcount'),
                      ('conjugated', 'order'))
    assert used.classes['nv'] <= full.classes['nv']
    assert GraphArena(dgs, (), ()).classes['nv'] == 1
    assert GraphArena(dgs, ('no_such_attribute',), ()).classes is None
    # class ids sit in front of every blob; equal ids <=> equal attributes
    node_t = np.dtype(dgs[0].node_t)
    seen = {}
    for k, g in enumerate(dgs):
        s = int(used.blob_start[k])
        cb = int(class_bytes(g.n_node, g.n_nz))
        assert cb % 16 == 0 and cb == used.class_bytes[k]
        ids = used.host[s - cb:s - cb + g.n_node]
        nodes = g.blob[g.offsets['node']:g.offsets['node']
                       + g.n_node * node_t.itemsize].view(node_t)
        for c, v in zip(ids, nodes):
            key = (int(v['aromatic']), int(v['atomic_number']),
                   int(v['hcount']))
            assert seen.setdefault(int(c), key) == key
    assert len(set(seen.values())) == len(seen) == used.classes['nv']
    # headers still address the blobs
    img = used.relocated(0)
    hdr = img[:64 * len(dgs)].view(np.dtype([
        ('n_node', np.int32), ('n_nz', np.int32), ('degree', np.uint32),
        ('node', np.uint32), ('rowptr', np.uint32), ('nz', np.uint32),
        ('edge', np.uint32), ('perm', np.uint32), ('hist', np.uint16, (16,))]))
    assert np.array_equal(hdr['degree'], used.blob_start
                          + dgs[0].offsets['degree'])
    # ... and carry the degree histogram the owner-computes solver lays its
    # row rectangles out from (graph.h: hist[d] = nodes with d nonzeros)
    for k, g in enumerate(dgs):
        want = np.bincount(np.diff(g.rowptr.astype(np.int64)), minlength=16)
        assert np.array_equal(hdr['hist'][k], want) and want.sum() == g.n_node


@pytest.mark.parametrize('symmetric', [True, False])
def test_reassembly_index_rebuilds_matrix_and_gradient_planes(symmetric):
    """ShardPlan.reassembly_index (the device-side gather + scatter of the
    multi-GPU step) against a direct construction: every rank's slab is
    [values | job-major gradient entries]."""
    rng = np.random.default_rng(4)
    nX, nY, world, nJ = 7, 7 if symmetric else 5, 3, 4
    if symmetric:
        i, j = np.triu_indices(nX)
    else:
        i, j = np.indices((nX, nY))
        i, j = i.ravel(), j.ravel() + nX
    n_node = rng.integers(3, 20, nX + (0 if symmetric else nY))
    n_nz = 2 * n_node
    value = rng.normal(size=len(i))
    grad = rng.normal(size=(len(i), nJ))
    plans = [ShardPlan(i, j, n_node, n_nz, nX, nY, symmetric, r, world)
             for r in range(world)]
    cap = plans[0].capacity
    slabs = []
    for sp in plans:                       # what each rank's solver writes
        slab = np.zeros(cap * (1 + nJ))
        slab[:len(sp.local)] = value[sp.local]
        slab[cap:cap + len(sp.local) * nJ] = grad[sp.local].ravel()
        slabs.append(slab)
    gathered = np.concatenate(slabs)
    src, dst = plans[0].reassembly_index(nJ)
    out = np.full((1 + nJ) * nX * nY, np.nan)
    out[dst] = gathered[src]
    K = out[:nX * nY].reshape(nX, nY, order='F')
    dK = out[nX * nY:].reshape(nJ, nY, nX).transpose(2, 1, 0)
    jy = j if symmetric else j - nX
    ref = np.full((nX, nY), np.nan)
    gref = np.full((nX, nY, nJ), np.nan)
    ref[i, jy], gref[i, jy] = value, grad
    if symmetric:
        ref[jy, i], gref[jy, i] = value, grad
    assert np.array_equal(K, ref) and np.array_equal(dK, gref)
    assert not np.isnan(K).any()
    # value-only form agrees with the host reassembly
    src1, dst1 = plans[0].reassembly_index(0)
    g1 = np.concatenate([s[:cap] for s in slabs])
    out1 = np.zeros(nX * nY)
    out1[dst1] = g1[src1]
    assert np.array_equal(out1.reshape(nX, nY, order='F'),
                          plans[0].assemble(g1))


def test_block_sharding_and_measured_rebalancing():
    """`partition_blocks`: contiguous blocks of the launch order, balanced on
    predicted job times plus a tail per solver variant a block touches, cuts
    snapped onto nearby variant boundaries; `ShardPlan.rebalanced`: when the
    ranks' measured times disagree with the prediction (here: one variant is
    really 40 % slower than the table says) the cuts move until the *real*
    times are balanced."""
    from graphdot_amd.kernel.marginalized._sharded import (
        ShardPlan, partition_blocks)
    rng = np.random.default_rng(5)
    n = 40000
    group = np.sort(rng.choice([3, 5, 6, 9], size=n, p=[.2, .4, .3, .1]))
    times = np.sort(rng.uniform(4, 12, n))[::-1] * (1 + 0.1 * group)
    order = np.arange(n)
    for world in (2, 4, 8):
        cuts = partition_blocks(times, group, world, tail=2000.0, snap=64)
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        cost = [times[a:b].sum() + 2000.0 * len(np.unique(group[a:b]))
                for a, b in cuts]
        assert max(cost) <= 1.03 * np.mean(cost)
    # a cut that falls next to a variant boundary moves onto it
    g2 = np.repeat([1, 2], [1000, 1010])
    cuts = partition_blocks(np.ones(2010), g2, 2, snap=64)
    assert cuts[0] == (0, 1000)
    # measured feedback
    ji = jj = np.zeros(n, dtype=np.int64)
    real = times * np.where(group == 5, 1.4, 1.0)      # what the GPU does
    world = 8
    sp = ShardPlan(ji, jj, np.ones(1), np.ones(1), 1, 1, False, 0, world,
                   launch_order=order, times=times, group=group, tail=2000.0,
                   snap=64)
    assert sp.mode == 'measured' and sorted(
        np.concatenate(sp.shards).tolist()) == list(range(n))

    def measure(plan):
        return np.array([real[s].sum() + 2000.0 * len(np.unique(group[s]))
                         for s in plan.shards])
    before = measure(sp)
    for _ in range(2):
        sp = sp.rebalanced(measure(sp))
    after = measure(sp)
    assert before.max() / before.mean() > 1.15
    assert after.max() / after.mean() < 1.04
    assert sorted(np.concatenate(sp.shards).tolist()) == list(range(n))


def test_measured_shard_plan_is_host_only_and_covers_every_job():
    """`_sharded.measured_shard_plan` (what ShardedStep builds its shards
    from) needs no device: the whole job list is laid out as one rank would
    launch it, every job gets the table time of its solver variant, the ranks
    take contiguous blocks of the launch order.  Every job lands in exactly one
    shard, every shard holds few solver variants, the predicted times are
    balanced, the plan is the same on every rank, and the merge map is the
    single-GPU one."""
    from graphdot_amd.kernel.marginalized._sharded import (
        measured_shard_plan, cost_table, variant_key)
    # (large enough that the pairs, not the launch tails, are what is dealt:
    # on a few thousand pairs the minimax plan rightly leaves ranks idle)
    G = cases.config3_graphs(400, seed=9)
    kn, ke, q = cases.config3_kernels()
    b = HIPBackend(real=np.float64)
    k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
    jobs = k._pairwise_jobs(len(G))
    table = cost_table()
    assert any(key.startswith('f64/C1/oc4_') for key in table)   # shipped
    plans = [measured_shard_plan(b, G, kn, ke, jobs, len(G), len(G),
                                 k.traits(symmetric=True), r, 4)
             for r in range(4)]
    sp = plans[0]
    assert sp.mode == 'measured'
    assert sorted(np.concatenate(sp.shards).tolist()) == list(range(len(jobs)))
    for other in plans[1:]:
        assert all(np.array_equal(a, c) for a, c in zip(sp.shards,
                                                        other.shards))
    assert max(sp.predicted) <= 1.15 * np.mean(sp.predicted)
    # the merge map is what a single GPU applies to the whole list
    dgs, ek, C, fields = b._graphs_and_kernels(G, kn, ke,
                                               k.traits(symmetric=True))
    arena = b._host_arena(dgs, fields)
    whole = b._partition(dgs, jobs, C, 0, b._global_tables(arena))
    assert sp.merge_map == whole.merge_map
    # a shard laid out with that map uses only variants the whole list uses
    _, used_all, _, _ = b._partition(dgs, jobs, C, 0,
                                     b._global_tables(arena))
    for s in sp.shards:
        _, used, order, launches = b._partition(
            dgs, np.ascontiguousarray(jobs[s]), C, 0,
            b._global_tables(arena), merge_map=sp.merge_map)
        assert set(used) <= set(used_all)
        assert sum(L['count'] for L in launches) == len(s)
    assert variant_key(b.variants[used_all[0]]).startswith('oc4_W1_')


def test_dense_graphs_classify_into_the_on_the_fly_variants():
    """Pairs whose rows have more terms than any register-slot variant holds
    (degree above 8: dense from_ase-like graphs) are assigned the on-the-fly
    variants (S = 0), for value and for value + gradient solves (two
    right-hand sides: twice the LDS for p) -- natively and in numpy alike."""
    from graphdot_amd.kernel.marginalized._backend_hip import OCVariant
    G = cases.tang2019_graphs(30, seed=1)
    kn, ke, q = cases.tang2019_kernels()
    i, j = np.triu_indices(len(G))
    got = {}
    for native in (True, False):
        b = HIPBackend(native=native)
        dgs = [b._register_graph(g) for g in G]
        assert max(d.max_degree for d in dgs) > 8
        c1, *_ = b.classify(i, j, dgs, 1)
        c2, *_ = b.classify(i, j, dgs, 2)
        got[native] = (c1, c2)
        v1 = [b.variants[c] for c in c1]
        assert sum(isinstance(v, OCVariant) and v.S == 0 for v in v1) \
            > 0.9 * len(v1)
        for c, a, bb in zip(c1, i, j):
            v = b.variants[c]
            if isinstance(v, OCVariant) and v.S == 0:
                assert dgs[a].n_node * dgs[bb].n_node <= 64 * v.W * v.R
        v2 = [b.variants[c] for c in c2]
        assert sum(isinstance(v, OCVariant) and v.S == 0 for v in v2) \
            > 0.9 * len(v2)
        for c, a, bb in zip(c2, i, j):
            v = b.variants[c]
            if isinstance(v, OCVariant) and v.S == 0:
                assert dgs[a].n_node * dgs[bb].n_node <= 64 * v.W * v.R
    assert np.array_equal(got[True][0], got[False][0])
    assert np.array_equal(got[True][1], got[False][1])


def test_packed_edge_records_are_offered_only_where_they_are_exact():
    """The dense product evaluates the edge microkernel on two records at once
    when the backend prints `edge2_t` (_backend_hip.py declstruct2): only for
    records made of 4-byte numbers without padding, and only for expressions
    that call nothing but what device/fmath.h overloads for pairs."""
    from graphdot_amd.kernel.marginalized._backend_hip import (
        HIPBackend, declstruct2, packable, packed_expression)
    import cases
    f4 = np.dtype([('weight', np.float32),
                   ('label', np.dtype([('length', np.float32),
                                       ('order', np.int32)], align=True))],
                  align=True)
    assert packable(f4)
    text = declstruct2(f4, 'edge2_t')
    assert text == ('struct edge2_t {graphdot::pk2<float32> weight; '
                    'struct edge2_t_label {graphdot::pk2<float32> length; '
                    'graphdot::pk2<int32> order;} label;};')
    for bad in (np.dtype([('a', np.float64)]),                    # 8-byte leaf
                np.dtype([('a', np.float32), ('b', np.int8)], align=True),
                np.dtype([('a', np.float32, (2,))]),              # array
                np.dtype([('a', np.bool_)]), np.dtype(np.float32)):
        assert not packable(bad), bad
    assert packed_expression(
        '((x1.weight * x2.weight) * (graphdot::exp(-0.5F*graphdot::ipow<2>('
        'x1.label.length - x2.label.length)/graphdot::ipow<2>(label.length.'
        'length_scale))))')
    assert packed_expression('(x1.a == x2.a ? 1.0f : a.h)')
    assert not packed_expression('graphdot::pow(x1.a, 2.5f)')
    assert not packed_expression('dotproduct(x1.a, x2.a)')
    assert not packed_expression('sqrtf(x1.a * x2.a)')
    # the rendered translation units: the weighted SquareExponential edge
    # kernel of the dense molecular set gets the packed record in float, not
    # in double
    kn, ke, q = cases.tang2019_kernels()
    G = cases.tang2019_graphs(3)
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend_hip import OCVariant
    from graphdot_amd.microkernel import TensorProduct, Product
    for real, want in ((np.float32, True), (np.float64, False)):
        be = HIPBackend(real=real)
        k = MarginalizedGraphKernel(kn, ke, q=q, backend=be)
        dg = [be._register_graph(g) for g in G]
        ke2 = TensorProduct(weight=Product(), label=ke)
        src = be.render_source(kn, ke2, k.p, dg[0].node_t, dg[0].edge_t,
                               [OCVariant(4, 0, 1, 0)], 1, weighted=True)
        assert ('using packed_edge_t = edge2_t;' in src) is want
        assert ('struct edge2_t {' in src) is want


def test_a_launch_is_sized_for_two_maxima_and_must_still_fit_the_lds():
    """Every pair gets a variant whose LDS regions hold it; the launch of a
    variant sizes its regions for the longest vector AND the largest graph
    image among its pairs, which can be two different pairs: the sum may
    exceed the 160 KB of a compute unit although every pair fits (found by
    scripts/fuzz_parity.py: invalid launches of the 16-wave double variant
    with LDS-resident slot values and of the 16-wave two-stage variant).
    `_fit_launches_into_lds` hands pairs that hold a maximum to the general
    solver until the launch fits."""
    from graphdot_amd.kernel.marginalized._backend_hip import (
        GENERAL, HIPBackend, LDS_LIMIT, NotOwnerComputes, OCVariant, Variant)
    be = HIPBackend(real=np.float64)
    g = be.variants.index(GENERAL)
    C, rsize = 1, 8
    for v in (OCVariant(16, 64, 3, 8), Variant(16, 64, 8)):
        k = be.variants.index(v)
        if isinstance(v, OCVariant):
            assert be.lds_slot_bytes(v, C) == 10 * 64 * 16 * 8

        def need(idx, vec, img):
            return be._launch_lds(v, C, vec[idx], vec[idx], img[idx], img[idx], 0)
        room = LDS_LIMIT - int(be.lds_bytes(v, C, 0, 0))      # vector + 2 images
        # pair 0: long vector, small images; pair 1: short vector, large
        # images; pair 2: small in both.  0 and 1 fit alone, not together.
        vec = np.array([(room - 2 * 9000) // rsize - 200, 1200, 900], np.int64)
        img = np.array([9000, (room - 1300 * rsize) // 2 - 200, 4000], np.int64)
        assert need([0], vec, img) <= LDS_LIMIT
        assert need([1], vec, img) <= LDS_LIMIT
        assert need([2], vec, img) <= LDS_LIMIT
        assert need([0, 1, 2], vec, img) > LDS_LIMIT
        choice = np.full(3, k, dtype=np.int64)
        out = be._fit_launches_into_lds(choice, C, vec, vec, img, img, 0, False)
        assert sorted(out.tolist()) == sorted([k, k, g])
        kept = np.flatnonzero(out == k)
        assert 2 in kept and need(kept, vec, img) <= LDS_LIMIT
        assert np.array_equal(choice, [k, k, k])            # (input untouched)
        # nothing to do: the same array comes back
        assert be._fit_launches_into_lds(out, C, vec, vec, img, img, 0,
                                         False) is out
        # calls that only the owner-computes solvers serve cannot fall back
        with pytest.raises(NotOwnerComputes):
            be._fit_launches_into_lds(choice, C, vec, vec, img, img, 0, True)


def test_workgroups_per_pair_of_the_streamed_solver():
    """`stream_parts`: one pair takes the whole chip, a list too long for a
    group per pair keeps one workgroup per pair (the hardware queue balances
    it), in between the pairs' costs decide -- a few groups that walk
    several pairs each beat one workgroup per pair when the largest pair
    would otherwise be the step."""
    from graphdot_amd.kernel.marginalized._backend_hip import (
        stream_parts, STREAM_MAX_PARTS)
    assert stream_parts(np.array([7.0]), 1, 256, 512) == min(256, STREAM_MAX_PARTS)
    assert stream_parts(None, 3, 256, 512) == 85         # (no costs: the first rule)
    rng = np.random.default_rng(0)
    n = rng.integers(150, 600, size=32)
    i, j = np.triu_indices(32)
    c = np.sort((n[i] * n[j]).astype(float))[::-1]
    assert stream_parts(c, len(c), 256, 512) == 1        # 528 pairs
    i, j = np.triu_indices(16)
    c = np.sort((n[i] * n[j]).astype(float))[::-1]
    m = stream_parts(c, len(c), 256, 512)                # 136 pairs
    assert 8 <= m <= 32, m
    # equal pairs, exactly one per compute unit: nothing to gain from sharing
    assert stream_parts(np.ones(256), 256, 256, 512) == 1
    # never more workgroups than the chip holds
    assert stream_parts(np.array([5.0, 4.0]), 2, 64, 128) <= 64


def test_lds_diagonals_of_the_six_batch_layout_on_the_host():
    """mgk_oc.h DLDS, host half: the double graph-level value kernel of the
    six-batch static layout keeps its Jacobi diagonals in LDS -- 2 R reals
    per lane more in the [Y] region, three waves per SIMD --, its nodal
    flavours, the value + gradient solver and the float build do not (two
    waves for the double ones).  (That host and
    device agree is a static_assert in every rendered kernel: the compile
    matrix checks it.)"""
    from graphdot_amd.kernel.marginalized._backend_hip import (
        HIPBackend, OCStatic)
    v6, v5 = OCStatic(16, 4, 4, 3, 1, 1), OCStatic(16, 4, 4, 1, 1)
    b = HIPBackend(real=np.float64)
    assert b.diagonals_in_lds(v6, 1) and not b.diagonals_in_lds(v6, 1, nodal=True)
    assert not b.diagonals_in_lds(v6, 2) and not b.diagonals_in_lds(v5, 1)
    extra = b.lds_bytes(v6, 1, 380, 800) - b.lds_bytes(v6, 1, 380, 800, nodal=True)
    assert extra == 2 * 6 * 64 * 8
    assert b._waves_without_lds_diagonals(v6, 1, False) == 3
    assert b._waves_without_lds_diagonals(v6, 1, True) == 2
    assert not HIPBackend(real=np.float32).diagonals_in_lds(v6, 1)
