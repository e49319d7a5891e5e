"""The distributed backend (`_sharded.distributed_backend`): two and EIGHT
ranks (one process each, sharing the single GPU of the test box through gloo)
evaluate the kernel through the ordinary API; every rank must end up with the
full matrix and gradient, equal to the single-process result."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    import torch                                   # noqa: F401  (first)
    import torch.distributed as dist
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._sharded import distributed_backend
    G = cases.config3_graphs(30, seed=6)
    knode, kedge, q = cases.config3_kernels()
    k = MarginalizedGraphKernel(knode, kedge, q=q,
                                backend=distributed_backend(device=0))
    K = k(G)
    K2, dK = k(G, eval_gradient=True)
    Kxy = k(G[:12], G[12:])
    d = k.diag(G)                                   # single-GPU path
    # a GPR on top: the regressor takes the all-gathered, reassembled matrix
    # and gradient planes from this rank's device (`device_gram` under the
    # distributed backend) -- no host arrays between solver and Cholesky
    from graphdot_amd.model.gaussian_process import GaussianProcessRegressor
    gpr = GaussianProcessRegressor(k, alpha=float(0.1 * d.mean()))
    gpr.X, gpr.y = G, np.cos(np.arange(len(G)))
    on_device = gpr._device_gramian(gpr._dense(), k, G, True)
    assert on_device is not None and on_device[0].is_cuda \
        and on_device[1].is_cuda
    assert on_device[1].shape == (len(G), len(G), len(k.theta))
    lml, glml = gpr.log_marginal_likelihood(eval_gradient=True)
    # the likelihood's gradient came from this rank's pairs and one
    # all-reduce: the gradient planes were never gathered
    step = k.backend.last_step
    assert step.n_grad == k.n_dims and not step.gather_gradient
    n_local = len(step.local_jobs)      # (the ranks' shares add up: parent)
    loo, gloo = gpr.squared_loocv_error(eval_gradient=True)
    assert k.backend.last_step.gather_gradient      # (needs whole planes)
    # masked targets: W covers the kept rows, the pairs are indexed in full
    y2 = [None if i % 7 == 3 else float(np.cos(i)) for i in range(len(G))]
    gpr2 = GaussianProcessRegressor(k, alpha=float(0.1 * d.mean()))
    gpr2.X, gpr2.y = G, y2
    lml2, glml2 = gpr2.log_marginal_likelihood(eval_gradient=True)
    # the overlapped form (off by default): the matrix from a value step of
    # its own, the value + gradient solvers detached on low-priority streams
    # while the factorisation is enqueued, joined before the contraction
    k.backend.overlap_min_ranks = 1
    assert k.backend.overlaps_dense_algebra() is True
    lml3, glml3 = gpr.log_marginal_likelihood(eval_gradient=True)
    k.backend.overlap_min_ranks = None
    assert k.backend.overlaps_dense_algebra() is False
    pending = k.backend.last_step
    assert not pending.gather_gradient and pending.n_grad == k.n_dims
    phases = step.phase_ms(steps=2)
    assert set(phases) == {'shard_ms', 'all_gather_ms', 'reassembly_ms'}
    assert all(v > 0 for v in phases.values())
    np.savez(os.path.join(tmp, f'rank{rank}.npz'), K=K, K2=K2, dK=dK,
             Kxy=Kxy, d=d, lml=lml, glml=glml, loo=loo, gloo=gloo,
             lml2=lml2, glml2=glml2, n_local=n_local, lml3=lml3,
             glml3=glml3)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_ranks_through_the_kernel_api(tmp_path, world):
    """Values, value + gradient, an X x Y block, `diag` and the Gaussian
    process step end to end on `world` ranks (eight: the rank count of the
    node BASELINE.json's configurations 4 and 5 name), every rank a process of
    its own, against one rank."""
    import torch.multiprocessing as mp
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    port = 29700 + (os.getpid() + 11 * world) % 200
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world,
             join=True)
    G = cases.config3_graphs(30, seed=6)
    knode, kedge, q = cases.config3_kernels()
    # (bit for bit: the same solver variant per pair -- launch merging is
    # decided on the whole job list and applied by every rank)
    from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
    k = MarginalizedGraphKernel(knode, kedge, q=q, backend=HIPBackend())
    K = k(G)
    K2, dK = k(G, eval_gradient=True)
    Kxy = k(G[:12], G[12:])
    d = k.diag(G)
    for rank in range(world):
        r = np.load(tmp_path / f'rank{rank}.npz')
        assert np.array_equal(r['K'], K)
        assert np.array_equal(r['K2'], K2) and np.array_equal(r['dK'], dK)
        assert np.array_equal(r['Kxy'], Kxy)
        assert np.array_equal(r['d'], d)
        assert np.array_equal(r['K'], r['K'].T)
    # the likelihood + gradient step of the regressor: two ranks through the
    # device-resident sharded path against one rank through the single-GPU
    # device path -- the same matrix bit for bit, hence the same objective
    from graphdot_amd.model.gaussian_process import GaussianProcessRegressor
    gpr = GaussianProcessRegressor(k, alpha=float(0.1 * d.mean()))
    gpr.X, gpr.y = G, np.cos(np.arange(len(G)))
    assert gpr._device_gramian(gpr._dense(), k, G, True) is not None
    lml, glml = gpr.log_marginal_likelihood(eval_gradient=True)
    loo, gloo = gpr.squared_loocv_error(eval_gradient=True)
    y2 = [None if i % 7 == 3 else float(np.cos(i)) for i in range(len(G))]
    gpr2 = GaussianProcessRegressor(k, alpha=float(0.1 * d.mean()))
    gpr2.X, gpr2.y = G, y2
    lml2, glml2 = gpr2.log_marginal_likelihood(eval_gradient=True)
    r0 = np.load(tmp_path / 'rank0.npz')
    # every unordered pair on exactly one rank
    assert sum(int(np.load(tmp_path / f'rank{r}.npz')['n_local'])
               for r in range(world)) == len(G) * (len(G) + 1) // 2
    for rank in range(world):
        r = np.load(tmp_path / f'rank{rank}.npz')
        # the same matrix bit for bit, hence the same factor and objective
        assert float(r['lml']) == lml and float(r['lml2']) == lml2
        assert float(r['loo']) == loo
        assert np.array_equal(r['gloo'], gloo)
        # the likelihood's gradient is summed pair shard by pair shard and
        # all-reduced: equal on every rank, round-off away from the
        # contraction of whole planes on one rank
        assert np.array_equal(r['glml'], r0['glml'])
        assert np.array_equal(r['glml2'], r0['glml2'])
        assert np.allclose(r['glml'], glml, rtol=1e-10, atol=1e-12 * np.abs(glml).max())
        assert np.allclose(r['glml2'], glml2, rtol=1e-10, atol=1e-12 * np.abs(glml2).max())
        # overlapped: the matrix comes from the VALUE solvers (other
        # instantiations than the value + gradient ones: round-off apart)
        assert float(r['lml3']) == pytest.approx(lml, rel=1e-6)
        assert np.allclose(r['glml3'], glml, rtol=1e-4,
                           atol=1e-6 * np.abs(glml).max())
    # ... and against the numpy kernel protocol (host arrays, float64
    # conversion on the host) to round-off
    slow = GaussianProcessRegressor(k, alpha=float(0.1 * d.mean()),
                                    kernel_options={'lmin': 0})
    slow.X, slow.y = G, np.cos(np.arange(len(G)))
    lml2, glml2 = slow.log_marginal_likelihood(eval_gradient=True)
    assert lml == pytest.approx(lml2, rel=1e-9)
    assert np.allclose(glml, glml2, rtol=1e-7)


def _rccl_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            device_id=torch.device('cuda', 0))
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._sharded import (
        distributed_backend, cuda_collective)
    assert cuda_collective()
    G = cases.config3_graphs(40, seed=6)
    knode, kedge, q = cases.config3_kernels()
    backend = distributed_backend(device=0, shard_single_rank=True)
    k = MarginalizedGraphKernel(knode, kedge, q=q, backend=backend)
    K = k(G)
    assert backend.last_step.on_device      # RCCL all-gather on device memory
    K2, dK = k(G, eval_gradient=True)
    # a second evaluation with other hyperparameters re-binds the cached step
    k2 = k.clone_with_theta(k.theta + 0.1)
    K3 = k2(G)
    Kxy = k(G[:12], G[12:])
    # the same all-gather through the C ABI (gd_comm_* / gd_all_gather):
    # the process group only carries the unique id
    abi = distributed_backend(device=0, shard_single_rank=True,
                              collective='rccl')
    k4 = MarginalizedGraphKernel(knode, kedge, q=q, backend=abi)
    K4, dK4 = k4(G, eval_gradient=True)
    assert abi.last_step.comm is not None
    # pipelined steps (two slab / gather buffers, solvers of step i + 1
    # overlapping the collective of step i): five back-to-back steps, one
    # download
    pipe = distributed_backend(device=0, shard_single_rank=True,
                               pipeline=True)
    k5 = MarginalizedGraphKernel(knode, kedge, q=q, backend=pipe)
    K5, dK5 = k5(G, eval_gradient=True)
    step = pipe.last_step
    assert step.depth == 2 and step.front is not None
    step.result.zero_()
    for _ in range(5):
        step.enqueue()
    K6 = np.empty(K5.size)
    dK6 = np.empty(dK5.size)
    step.download(K6, dK6)
    n = len(G)
    K6 = K6.reshape(n, n, order='F')
    dK6 = dK6.reshape(n, n, -1, order='F')
    np.savez(os.path.join(tmp, 'rccl.npz'), K=K, K2=K2, dK=dK, Kxy=Kxy, K3=K3,
             theta=k2.theta, K4=K4, dK4=dK4, K5=K5, dK5=dK5, K6=K6, dK6=dK6)
    dist.destroy_process_group()


def test_rccl_process_group_of_one_rank(tmp_path):
    """The device-resident sharded step (`ShardedStep`: kernels write the
    packed slab into the all-gather input, RCCL `all_gather_into_tensor` on
    device memory, reassembly on the device, one download) on a "nccl"
    process group of size 1 -- everything of the N-GPU path except a second
    rank -- reproduces the single-GPU results bit for bit."""
    import torch.multiprocessing as mp
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    port = 29950 + os.getpid() % 40
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / 'rccl.npz')
    G = cases.config3_graphs(40, seed=6)
    knode, kedge, q = cases.config3_kernels()
    from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
    k = MarginalizedGraphKernel(knode, kedge, q=q, backend=HIPBackend())
    K2, dK = k(G, eval_gradient=True)
    assert np.array_equal(r['K'], k(G))
    assert np.array_equal(r['K2'], K2) and np.array_equal(r['dK'], dK)
    assert np.array_equal(r['Kxy'], k(G[:12], G[12:]))
    assert np.array_equal(r['K4'], K2) and np.array_equal(r['dK4'], dK)
    assert np.array_equal(r['K5'], K2) and np.array_equal(r['dK5'], dK)
    assert np.array_equal(r['K6'], K2) and np.array_equal(r['dK6'], dK)
    k.theta = r['theta']
    assert np.array_equal(r['K3'], k(G))


def test_bench_sharded_step_single_rank():
    """bench.py's multi-GPU step on one rank (`--sharded`: nccl group of
    size 1): the JSON line carries the oracle check of the reassembled
    matrix."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(29990 + os.getpid() % 9),
               MASTER_ADDR='127.0.0.1')
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--sharded',
         '--graphs', '150', '--steps', '3', '--warmup', '1',
         '--no-cpu-baseline', '--dtype', 'f32', '--gradient'],
        capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])   # JSON comes last
    chk = line['sharded_check']
    assert chk['collective'] == 'nccl(RCCL)' and chk['symmetric']
    assert chk['max_rel_diff_vs_oracle'] < 1e-5
    assert line['config']['parallelism'] == 'pair-sharded x1'


def test_bench_spawns_its_ranks():
    """`python bench.py --gpus 2` without a launcher starts its two ranks
    itself (before anything touches the GPU in the parent) instead of running
    one rank and printing n_gpus = 1; the ranks share the test GPU (gloo).
    The same for the GPR step of configuration 5, whose kernel matrix stays
    on the device on every rank."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT',
                        'MASTER_ADDR')}
    common = ['--gpus', '2', '--graphs', '120', '--steps', '2', '--warmup',
              '1', '--dtype', 'f32', '--share-devices']
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), *common,
         '--no-cpu-baseline'],
        capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # ONE line on stdout: library banners (gloo's connection report) go to
    # stderr
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:2000]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2
    assert line['config']['parallelism'] == 'pair-sharded x2'
    assert line['sharded_check']['ranks'] == 2
    assert line['sharded_check']['max_rel_diff_vs_oracle'] < 1e-5
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), *common, '--gpr'],
        capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['device_resident_kernel_matrix']
    # a mismatch between --gpus and the launcher's world size is an error
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'],
        capture_output=True, text=True, env=dict(env, WORLD_SIZE='2'),
        timeout=120)
    assert r.returncode != 0


def _device_count():
    import torch
    return torch.cuda.device_count()


def test_bench_refuses_to_share_devices_silently():
    """`bench.py --gpus N` on a node that shows fewer than N GPUs leaves with
    a non-zero code on every rank and prints no JSON line: ranks sharing a
    device over gloo never pass for an N-GPU measurement (that mode has to be
    asked for with --share-devices)."""
    import subprocess
    n = _device_count() + 1
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT',
                        'MASTER_ADDR')}
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n),
         '--graphs', '60', '--steps', '1', '--warmup', '1', '--dtype', 'f32',
         '--no-cpu-baseline'],
        capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ''
    assert 'distinct GPUs' in r.stderr


def _alternating_worker(rank, world, port, tmp, backend_name, n_dev):
    import torch
    import torch.distributed as dist
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    device = rank % n_dev
    torch.cuda.set_device(device)
    if backend_name == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', device))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._sharded import (
        distributed_backend, cuda_collective)
    assert cuda_collective() == (backend_name == 'nccl')
    A = cases.config3_graphs(48, seed=31)
    B = cases.config3_graphs(48, seed=32)           # same size, other graphs
    knode, kedge, q = cases.config3_kernels()
    out = {}
    for name, kw in (('torch', {}), ('rccl', {'collective': 'rccl'})):
        if name == 'rccl' and backend_name != 'nccl':
            continue
        backend = distributed_backend(device=device, **kw)
        # (measured re-balancing on these small job lists too: it replaces
        # the cached plan of the evaluation it tuned -- and only that one)
        backend.rebalance_min_jobs = 1
        k = MarginalizedGraphKernel(knode, kedge, q=q, backend=backend)
        out[f'{name}_KA'] = k(A)
        assert backend.last_step.world == world
        assert backend.last_step.on_device == (backend_name == 'nccl')
        out[f'{name}_KB'] = k(B)
        out[f'{name}_KA2'], out[f'{name}_dKA'] = k(A, eval_gradient=True)
        out[f'{name}_KB2'], out[f'{name}_dKB'] = k(B, eval_gradient=True)
        out[f'{name}_KA3'] = k(A)                   # cached plans, third round
        out[f'{name}_KB3'] = k(B)
        out[f'{name}_KA4'], out[f'{name}_dKA4'] = k(A, eval_gradient=True)
    np.savez(os.path.join(tmp, f'alt{rank}.npz'), **out)
    dist.destroy_process_group()


def _check_alternating(tmp_path, names):
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
    A = cases.config3_graphs(48, seed=31)
    B = cases.config3_graphs(48, seed=32)
    knode, kedge, q = cases.config3_kernels()
    k = MarginalizedGraphKernel(knode, kedge, q=q, backend=HIPBackend())
    KA, KB = k(A), k(B)
    KA2, dKA = k(A, eval_gradient=True)
    KB2, dKB = k(B, eval_gradient=True)
    assert not np.allclose(KA, KB)
    for rank in range(2):
        r = np.load(tmp_path / f'alt{rank}.npz')
        for n in names:
            assert np.array_equal(r[f'{n}_KA'], KA)
            assert np.array_equal(r[f'{n}_KB'], KB)
            assert np.array_equal(r[f'{n}_KA2'], KA2)
            assert np.array_equal(r[f'{n}_dKA'], dKA)
            assert np.array_equal(r[f'{n}_KB2'], KB2)
            assert np.array_equal(r[f'{n}_dKB'], dKB)
            assert np.array_equal(r[f'{n}_KA3'], KA)
            assert np.array_equal(r[f'{n}_KB3'], KB)
            assert np.array_equal(r[f'{n}_KA4'], KA2)
            assert np.array_equal(r[f'{n}_dKA4'], dKA)


def test_two_ranks_alternating_graph_sets_and_gradient_calls(tmp_path):
    """Two graph sets of EQUAL size evaluated in turn, values and value +
    gradient in turn, on two ranks with measured re-balancing on: the job
    list of an n x n matrix is one cached object shared by every set of n
    graphs, so the tuned shard plan of one evaluation must not replace the
    plan (and with it the cached step, bound to other graphs) of another.
    Every result bit-equal to the single-GPU one."""
    import torch.multiprocessing as mp
    port = 29300 + os.getpid() % 200
    mp.spawn(_alternating_worker, args=(2, port, str(tmp_path), 'gloo', 1),
             nprocs=2, join=True)
    _check_alternating(tmp_path, ['torch'])


@pytest.mark.skipif(_device_count() < 2, reason='needs two GPUs')
def test_two_ranks_on_two_devices_over_rccl(tmp_path):
    """The multi-device branch proper (runs wherever two GPUs are visible):
    two ranks on two devices, "nccl" process group (RCCL over xGMI), through
    `distributed_backend()` with torch's all-gather and with
    `collective='rccl'` (gd_comm_unique_id / gd_comm_init_rank /
    gd_all_gather of the C ABI, per-rank gd_init(local_rank)); alternating
    graph sets, values and gradients; bit-equal to one rank."""
    import torch.multiprocessing as mp
    port = 29100 + os.getpid() % 200
    mp.spawn(_alternating_worker, args=(2, port, str(tmp_path), 'nccl', 2),
             nprocs=2, join=True)
    _check_alternating(tmp_path, ['torch', 'rccl'])


@pytest.mark.skipif(_device_count() < 2, reason='needs two GPUs')
def test_bench_two_gpus_sharded():
    """`bench.py --gpus 2` on two real devices: RCCL, two ranks, the
    reassembled matrix against the oracle; and the GPR step on two ranks."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT',
                        'MASTER_ADDR')}
    for extra in (['--no-cpu-baseline'], ['--gradient', '--no-cpu-baseline'],
                  ['--gpr']):
        r = subprocess.run(
            [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
             '--graphs', '200', '--steps', '3', '--warmup', '1', *extra],
            capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line['n_gpus'] == 2
        if '--gpr' in extra:
            assert line['collective'] == 'nccl(RCCL)'
            assert line['device_resident_kernel_matrix']
        else:
            chk = line['sharded_check']
            assert chk['collective'] == 'nccl(RCCL)' and chk['ranks'] == 2
            # (double arithmetic at the reference's stopping rule, 1e-8 N)
            assert chk['max_rel_diff_vs_oracle'] < 1e-6
