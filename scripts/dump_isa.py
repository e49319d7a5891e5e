#!/usr/bin/env python3
"""Dump the gfx950 ISA of one solver variant built for the bench workload
(QM7-like TensorProduct kernels) -- for instruction-count work on the CG loop.

    python scripts/dump_isa.py W S R [C] [--f64] [--oc=D] [--layout=16x4x4x1] [--tab|--tab=2] [--config2|--tang] > out.s
"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                  # noqa: E402
import cases                                                        # noqa: E402
from graphdot_amd.hip import jit                                    # noqa: E402
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel  # noqa
from graphdot_amd.kernel.marginalized._backend_hip import (         # noqa: E402
    HIPBackend, Variant, OCVariant, OCStatic)

args = [a for a in sys.argv[1:] if not a.startswith('--')]
W, S, R = map(int, args[:3])
C = int(args[3]) if len(args) > 3 else 1
real = np.float64 if '--f64' in sys.argv else np.float32
backend = HIPBackend(real=real)
if '--config2' in sys.argv:      # weighted graphs, continuous edge labels
    kn, ke, q = cases.config2b_kernels()
    G = cases.config2_graphs(8, seed=0)
elif '--tang' in sys.argv:       # dense molecular graphs (on-the-fly solvers)
    kn, ke, q = cases.tang2019_kernels()
    G = cases.tang2019_graphs(8)
else:
    kn, ke, q = cases.config3_kernels()
    G = cases.config3_graphs(8)
k = MarginalizedGraphKernel(kn, ke, q=q, backend=backend)
dgs = [backend._register_graph(g) for g in G]
node_t, edge_t = dgs[0].node_t, dgs[0].edge_t
ke2 = ke
if dgs[0].weighted:      # (as HIPBackend._graphs_and_kernels wraps it)
    from graphdot_amd.microkernel import TensorProduct, Product
    ke2 = TensorProduct(weight=Product(), label=ke)
oc = [int(a.split('=')[1]) for a in sys.argv if a.startswith('--oc=')]
lay = [a.split('=')[1] for a in sys.argv if a.startswith('--layout=')]
variant = OCVariant(W, S, R, oc[0]) if oc else Variant(W, S, R)
if lay:       # --layout=16x4x4x1 (W S R are then ignored)
    variant = OCStatic(*map(int, lay[0].split('x')))
src = backend.render_source(kn, ke2, k.p, node_t, edge_t, [variant], C,
                            tab=2 if '--tab=2' in sys.argv else '--tab' in sys.argv,
                            weighted=dgs[0].weighted)
path = f'/tmp/_dump_isa_{W}_{S}_{R}_{C}_{int(real is np.float64)}.hip'
open(path, 'w').write(src)
flags = [f for f in jit.BASE_FLAGS if f != '--genco'] + \
    os.environ.get('GD_HIPCC_EXTRA', '').split()
cmd = [jit.HIPCC, *flags, '--cuda-device-only', '-S', f'-I{jit.DEVICE_INCLUDE}',
       path, '-o', '-']
sys.stdout.write(subprocess.run(cmd, capture_output=True, text=True,
                                check=True).stdout)
