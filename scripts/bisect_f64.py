import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), '..'), os.path.join(os.path.dirname(__file__), '..', 'tests')]
import numpy as np
import test_parity_gpu as t
from oracle import mgk
from graphdot_amd.microkernel import *
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
G = t._feature_graphs(real=np.float64)
b = HIPBackend(real=np.float64)
combos = {
 'normdot+SE': (TensorProduct(fp=Normalize(DotProduct()), category=KroneckerDelta(0.4)), TensorProduct(length=SquareExponential(1.0))),
 'KD+RQ(.8,.7)': (TensorProduct(category=KroneckerDelta(0.4)), TensorProduct(length=RationalQuadratic(0.8, 0.7))),
 'KD+RQ(1,1.5)': (TensorProduct(category=KroneckerDelta(0.4)), TensorProduct(length=RationalQuadratic(1.0, 1.5))),
 'KD+SE(.8)': (TensorProduct(category=KroneckerDelta(0.4)), TensorProduct(length=SquareExponential(0.8))),
 'RQnode(.8,.7)+SE': (TensorProduct(radius=RationalQuadratic(0.8, 0.7), category=KroneckerDelta(0.5)), TensorProduct(length=SquareExponential(1.0))),
}
for name, (kn, ke) in combos.items():
    k = MarginalizedGraphKernel(kn, ke, q=0.05, backend=b, ftol=1e-13)
    K = k(G)
    Ko = mgk.gram(G, kn, ke, q=0.05, tol=1e-13)
    print(name, np.max(np.abs(K / Ko - 1)))
