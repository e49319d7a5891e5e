cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python scripts/soak.py 2>&1 | tail -6
timeout 900 python scripts/stale_cache_stress.py 150 2>&1 | tail -4
timeout 600 python - <<'PY'
import sys, gc, time
sys.path[:0] = ['.', 'tests']
import numpy as np, cases
from graphdot_amd.hip import runtime
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
G = cases.config3_graphs(1000)
kn, ke, q = cases.config3_kernels()
k = MarginalizedGraphKernel(kn, ke, q=q, backend=HIPBackend(real=np.float64))
ref = k(G)
held = []
for it in range(40):
    K = k(G) if it % 4 else k(G, eval_gradient=True)[0]
    assert np.array_equal(K, ref) or it % 4 == 0
    if it % 5 == 0:
        held.append(K)            # results kept alive: the pool must allocate
print('pinned live blocks', len(runtime._pinned_live), 'idle bytes', runtime._pinned_idle >> 20, 'MB; is_pinned(K):', runtime.is_pinned(K.ravel(order="K")))
del held, K
gc.collect()
print('after release: live', len(runtime._pinned_live), 'idle MB', runtime._pinned_idle >> 20)
PY
