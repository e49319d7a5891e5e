#!/usr/bin/env python
"""Build-container-only check of the drop-in seam (needs /root/reference):
hand a HIPBackend instance to the *reference's* MarginalizedGraphKernel,
feed it the reference's own Graph / microkernel objects and run everything
that does not need a device (graph packing, code generation, hipcc JIT,
job partitioning).  Used while writing INTEGRATION.md; not part of the
test-suite (the reference cannot travel to the GPU box)."""
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
    from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend

    # the reference's factory accepts instances of ITS Backend ABC
    class DropIn(HIPBackend, RefBackend):
        pass

    ns = load_test_oracle()
    backend = DropIn()
    for name, case in ns['case_dict'].items():
        k = MarginalizedGraphKernel(case['knode'], case['kedge'], q=0.05,
                                    backend=backend)
        assert k.backend is backend
        G = case['graphs']
        i, j = np.triu_indices(len(G))
        jobs = np.column_stack((i, j)).astype(np.uint32).ravel().view(
            np.dtype([('i', np.uint32), ('j', np.uint32)]))
        for eg in (False, True):
            paths = backend.precompile(
                G, k.node_kernel, k.edge_kernel, k.p, jobs,
                k.traits(symmetric=True, eval_gradient=eg))
            print(name, 'gradient' if eg else 'value', '->',
                  [os.path.basename(p) for p in paths])
    print('drop-in seam ok: reference kernel object + HIPBackend')


if __name__ == '__main__':
    main()
