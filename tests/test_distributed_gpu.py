"""The distributed backend (`_sharded.distributed_backend`): two ranks (one
process each, sharing the single GPU of the test box through gloo) evaluate
the kernel through the ordinary API; every rank must end up with the full
matrix and gradient, equal to the single-process result."""
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
    # a GPR on top: the zero-copy device path steps aside, the likelihood
    # goes through the sharded __call__ and is the same on every rank
    from graphdot_amd.model.gaussian_process import GaussianProcessRegressor
    gpr = GaussianProcessRegressor(k, alpha=float(0.1 * d.mean()))
    gpr.X, gpr.y = G, np.cos(np.arange(len(G)))
    assert gpr._device_gramian(gpr._dense(), k, G, True) is None
    lml, glml = gpr.log_marginal_likelihood(eval_gradient=True)
    np.savez(os.path.join(tmp, f'rank{rank}.npz'), K=K, K2=K2, dK=dK,
             Kxy=Kxy, d=d, lml=lml, glml=glml)
    dist.destroy_process_group()


def test_two_ranks_through_the_kernel_api(tmp_path):
    import torch.multiprocessing as mp
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    port = 29700 + os.getpid() % 200
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    G = cases.config3_graphs(30, seed=6)
    knode, kedge, q = cases.config3_kernels()
    k = MarginalizedGraphKernel(knode, kedge, q=q)
    K = k(G)
    K2, dK = k(G, eval_gradient=True)
    Kxy = k(G[:12], G[12:])
    d = k.diag(G)
    for rank in range(2):
        r = np.load(tmp_path / f'rank{rank}.npz')
        assert np.array_equal(r['K'], K)
        assert np.array_equal(r['K2'], K2) and np.array_equal(r['dK'], dK)
        assert np.array_equal(r['Kxy'], Kxy)
        assert np.array_equal(r['d'], d)
        assert np.array_equal(r['K'], r['K'].T)
    from graphdot_amd.model.gaussian_process import GaussianProcessRegressor
    gpr = GaussianProcessRegressor(k, alpha=float(0.1 * d.mean()),
                                   kernel_options={'lmin': 0})   # numpy path
    gpr.X, gpr.y = G, np.cos(np.arange(len(G)))
    lml, glml = gpr.log_marginal_likelihood(eval_gradient=True)
    r0, r1 = (np.load(tmp_path / f'rank{r}.npz') for r in range(2))
    assert float(r0['lml']) == float(r1['lml']) == pytest.approx(lml, rel=1e-9)
    assert np.allclose(r0['glml'], glml, rtol=1e-7)
