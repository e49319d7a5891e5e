#!/usr/bin/env python
"""Check of the drop-in seam against the REFERENCE's own caller (needs
/root/reference: build container only; `tests/test_host_model.py::
test_reference_kernel_object_drives_the_hip_backend` runs this script in a
process of its own when the reference is there and skips otherwise).

A HIPBackend instance is handed to the *reference's* MarginalizedGraphKernel
(`graphdot/kernel/marginalized/_kernel.py:60-73`, `_backend_factory.py:7-9`)
together with the reference's own Graph / microkernel objects, and the
reference's `__call__` (`_kernel.py:114-264`, backend call `:224-242`) and
`diag` (`:266-408`, backend call `:363-381`) construct the arguments --
allocators `backend.array / zeros / empty`, job list, `starts`, output
buffers, traits.  `HIPBackend.__call__` is intercepted at its last host-only
point: everything of `prepare` that needs no device runs on those arguments
(graph packing from the reference's Graph objects, solver variants and launch
order, code generation from the reference's microkernels, hipcc JIT, the
kernel-argument struct filled from the reference's hyperparameter states) for
value, value + gradient, nodal, nodal + gradient, X x Y blocks and the three
`diag` modes; no launch."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
from make_golden import install_shims, load_test_oracle  # noqa: E402


def main():
    install_shims()
    sys.path.insert(0, '/root/reference')
    from graphdot.kernel.marginalized import MarginalizedGraphKernel
    from graphdot.kernel.marginalized._backend import Backend as RefBackend
    from graphdot_amd.kernel.marginalized._backend_hip import (
        HIPBackend, pack_theta)
    from graphdot_amd.hip import jit

    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
    seen = []

    # the reference's factory accepts instances of ITS Backend ABC
    class DropIn(HIPBackend, RefBackend):
        def __call__(self, graphs, node_kernel, edge_kernel, p, q, eps, ftol,
                     gtol, jobs, starts, gramian, gradient, nX, nY, nJ,
                     traits, timer):
            # -- the contract of the seam (reference _kernel.py:172-242) ----
            assert isinstance(jobs, np.ndarray) and jobs.dtype == job_t
            assert isinstance(starts, np.ndarray) and starts.dtype == np.uint32
            assert gramian.dtype == np.float32 and gramian.ndim == 1
            assert gramian.size == nX * nY
            if traits.eval_gradient is True:
                assert gradient.dtype == np.float32
                assert gradient.size == nX * nY * nJ
            else:
                assert gradient is None
            assert jobs['i'].max() < len(graphs) and jobs['j'].max() < len(graphs)
            # -- host-only half of HIPBackend.prepare --------------------------
            ngrad = traits.eval_gradient is True and traits.nodal is True
            if traits.eval_gradient is True and traits.nodal == 'block':
                traits = traits._replace(eval_gradient=False)
            dgraphs, ek, C, fields = self._graphs_and_kernels(
                graphs, node_kernel, edge_kernel, traits, timer, ngrad)
            arena = self._host_arena(dgraphs, fields)
            tab_bytes = 0 if ngrad else self._table_bytes(arena)
            gtab = self._global_tables(arena) and not ngrad
            jobs_, used, order_all, launches = self._partition(
                dgraphs, jobs, C, tab_bytes, gtab, oc_only=ngrad)
            assert sorted(order_all.tolist()) == list(range(len(jobs)))
            assert sum(L['count'] for L in launches) == len(jobs)
            sources = self._sources(used, node_kernel, ek, p, dgraphs, C,
                                    traits.nodal is not False, tab_bytes > 0,
                                    gtab, ngrad)
            paths = jit.compile_many(list(sources.values()), self.hipcc_extra)
            # kernel arguments from the reference's hyperparameter objects
            pd = self._params_dtype(node_kernel, ek, p)
            base = np.zeros((), dtype=pd)
            for field, obj in (('node_kernel', node_kernel),
                               ('edge_kernel', ek), ('p_start', p)):
                _, val = pack_theta(obj, self.real)
                if val is not None:
                    base[field] = val
            base['q'], base['ftol'] = q, ftol
            seen.append(dict(traits=traits, n_jobs=len(jobs), nX=nX, nY=nY,
                             nJ=nJ, code_objects=len(paths),
                             launches=len(launches)))
            gramian[:] = 1.0                     # (no launch: placeholders)
            if gradient is not None:
                gradient[:] = 0.0

    ns = load_test_oracle()
    backend = DropIn()
    for name, case in ns['case_dict'].items():
        k = MarginalizedGraphKernel(case['knode'], case['kedge'], q=0.05,
                                    backend=backend)
        assert k.backend is backend
        G = case['graphs']
        n = len(G)
        nn = sum(len(g.nodes) for g in G)
        n_active = int(np.count_nonzero(k.active_theta_mask))
        before = len(seen)
        K = k(G)
        assert K.shape == (n, n) and seen[-1]['n_jobs'] == n * (n + 1) // 2
        K, dK = k(G, eval_gradient=True)
        assert dK.shape == (n, n, n_active) and seen[-1]['nJ'] == k.n_dims
        assert k(G, nodal=True).shape == (nn, nn)
        K, dK = k(G, nodal=True, eval_gradient=True)
        assert dK.shape == (nn, nn, n_active)
        assert k(G[:1], G[1:]).shape == (1, n - 1)
        assert seen[-1]['n_jobs'] == n - 1 and not seen[-1]['traits'].symmetric
        assert k(G, lmin=1).shape == (n, n) and seen[-1]['traits'].lmin == 1
        assert k.diag(G).shape == (n,) and seen[-1]['traits'].diagonal
        d, dd = k.diag(G, eval_gradient=True)
        assert dd.shape == (n, n_active)
        assert k.diag(G, nodal=True).shape == (nn,)
        blocks = k.diag(G, nodal='block')
        assert [b.shape for b in blocks] == [(len(g.nodes),) * 2 for g in G]
        print(name, '->', len(seen) - before, 'backend calls,',
              sum(s['code_objects'] for s in seen[before:]), 'code objects')
    print('drop-in seam ok: reference kernel object + HIPBackend')


if __name__ == '__main__':
    main()
