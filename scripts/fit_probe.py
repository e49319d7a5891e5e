"""Per-evaluation timings of the L-BFGS-B fit of bench.py --gpr --fit (kernel
part, dense part, pseudo-inverse fallbacks)."""
import sys, time, warnings
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np, cases
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
from graphdot_amd.model.gaussian_process import GaussianProcessRegressor
n=1000
G=cases.config3_graphs(n); kn,ke,q=cases.config3_fit_kernels()
y=cases.synthetic_energies(G)
b=HIPBackend(real=np.float64)
k=MarginalizedGraphKernel(kn,ke,q=q,q_bounds=(1e-3,0.5),backend=b)
d=k.diag(G)
gpr=GaussianProcessRegressor(k, alpha=1e-2, normalize_y=True, optimizer=True)
log=[]
orig=gpr.log_marginal_likelihood
def wrapped(*a,**kw):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        t=time.perf_counter(); out=orig(*a,**kw); dt=time.perf_counter()-t
    log.append((dt*1e3, gpr.last_timing['kernel']*1e3, gpr.last_timing['linalg']*1e3, len(w), float(out[0]) if isinstance(out,tuple) else float(out)))
    return out
gpr.log_marginal_likelihood=wrapped
t=time.perf_counter(); gpr.fit(G,y); print('fit s', time.perf_counter()-t)
for r in log: print('%.2f kernel %.2f dense %.2f warnings %d value %.6g'%r)
