"""Build-time compile matrix: every microkernel family x output mode x
arithmetic x solver family is rendered and compiled by hipcc for gfx950 -- no
launch, no GPU.

Why: round 4's defect (iv) was a JIT *assembly* failure (``v_writelane_b32 ...
illegal VGPR to SGPR copy`` in the nodal-gradient kernels of rational-quadratic
composites) that surfaced at run time on the GPU box, because the test suite
compiled one composite only.  The generated code of a microkernel changes the
register allocation of the solver around it, so every family of the
reference's grammar (graphdot/microkernel/_base.py:16-730: Constant,
KroneckerDelta, SquareExponential, RationalQuadratic, Normalize, DotProduct,
Convolution, TensorProduct / Additive composites and the ``+ * **``
operators) is compiled into every solver mode here.

``sources()`` yields (label, hipcc flags, translation unit);
``__graft_entry__.build()`` compiles them into the JIT cache,
``tests/test_compile_matrix.py`` checks that every one compiles (instantly
when the cache is warm).
"""
import numpy as np


def families():
    """(name, graphs, node kernel, edge kernel, label-class tables?)"""
    import cases
    from graphdot_amd.microkernel import (
        Constant, KroneckerDelta, SquareExponential, RationalQuadratic,
        DotProduct, Normalize, Convolution, TensorProduct, Additive)
    out = []
    kn, ke, _ = cases.config1_kernels()             # Constant x Constant
    out.append(('constant', cases.config1_graphs(), kn, ke))
    kn, ke, _ = cases.config2a_kernels()            # KroneckerDelta both
    out.append(('kronecker', cases.nlw_example_graphs(), kn, ke))
    kn, ke, _ = cases.config2b_kernels()            # delta x square exponential, weighted
    out.append(('delta_x_sqexp', cases.config2_graphs(4, seed=3), kn, ke))
    kn, ke, _ = cases.config3_kernels()             # tensor products over label classes
    out.append(('molecular_tables', cases.config3_graphs(6, seed=5), kn, ke))
    F = cases.feature_graphs(n_graphs=3)
    out.append(('rational_quadratic', F,
                TensorProduct(radius=RationalQuadratic(1.0, 1.5),
                              category=KroneckerDelta(0.5)),
                TensorProduct(length=RationalQuadratic(0.8, 0.7))))
    out.append(('normalized_dot_product', F,
                TensorProduct(fp=Normalize(DotProduct()),
                              category=KroneckerDelta(0.4)),
                TensorProduct(length=SquareExponential(1.0))))
    out.append(('operators', F,
                TensorProduct(radius=SquareExponential(0.7),
                              category=KroneckerDelta(0.5)) ** 1.5,
                (TensorProduct(length=SquareExponential(1.2)) * 0.6 + 0.4)
                ** 2.5))
    out.append(('additive', F,
                Additive(radius=SquareExponential(0.7) * Constant(0.5),
                         category=KroneckerDelta(0.5)),
                Additive(length=RationalQuadratic(0.8, 0.7))))
    R = cases.config3_graphs(5, seed=21, ring_list=True)
    out.append(('convolution', R,
                TensorProduct(atomic_number=KroneckerDelta(0.5),
                              ring_list=Convolution(KroneckerDelta(0.6))),
                TensorProduct(order=SquareExponential(0.5))))
    return out


#: (C, nodal, ngrad, maximin)
MODES = {
    'value': (1, False, False, False),
    'gradient': (2, False, False, False),
    'nodal': (1, True, False, False),
    'ngrad': (1, True, True, False),
    'maximin': (1, True, False, True),
    'maximin_ngrad': (1, True, True, True),
}


def shapes(full):
    """Solver shapes: static one-wave layouts, dynamic layouts with 1, 4, 8
    and 16 waves (the 16-wave double ones park slot values in LDS and pack
    their gather addresses), on-the-fly, two-stage, general.  `full`: the
    whole list; otherwise one shape per solver family."""
    from graphdot_amd.kernel.marginalized._backend_hip import (
        OCVariant, OCStatic, Variant, GENERAL)
    oc = [OCStatic(16), OCVariant(4, 32, 3, 8), OCVariant(4, 0, 1, 0)]
    two_stage = [Variant(1, 8, 2)]
    if full:
        oc += [OCStatic(16, 4, 4, 1), OCVariant(1, 12, 2, 4),
               OCVariant(8, 64, 4, 8), OCVariant(16, 40, 2, 8),
               OCVariant(16, 0, 2, 0)]
        two_stage += [Variant(16, 16, 2)]
    return oc, two_stage, [GENERAL]


def sources(select=None):
    """Yield (label, flags, source) over the matrix.  The two BASELINE
    families (direct evaluation: `delta_x_sqexp`, label-class tables:
    `molecular_tables`) take every solver shape, the others one shape per
    solver family."""
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend_hip import (
        HIPBackend, OCVariant, GENERAL, MFMA, STREAM)
    for fname, graphs, knode, kedge in families():
        if select and fname not in select:
            continue
        full = fname in ('delta_x_sqexp', 'molecular_tables')
        oc, two_stage, general = shapes(full)
        for real in (np.float32, np.float64):
            backend = HIPBackend(real=real)
            k = MarginalizedGraphKernel(knode, kedge, q=0.05, backend=backend)
            rname = 'f64' if real is np.float64 else 'f32'
            for mname, (C, nodal, ngrad, maximin) in MODES.items():
                traits = k.traits(symmetric=not nodal, nodal=nodal,
                                  eval_gradient=C == 2)
                dgraphs, ek, _, fields = backend._graphs_and_kernels(
                    graphs, knode, kedge, traits, None, ngrad)
                arena = backend._host_arena(dgraphs, fields)
                # (the nodal-gradient and maximin solvers evaluate the
                # microkernels directly: HIPBackend._layout)
                gtab = backend._global_tables(arena) and not ngrad \
                    and not maximin
                todo = [(v, gtab) for v in oc
                        if not (v.S == 0 and C == 2 and nodal)]
                if not (ngrad or maximin):
                    todo += [(v, False) for v in two_stage + general]
                # the streamed solver of large pairs: values (any output
                # mode) and graph-level value + gradient
                if not (ngrad or maximin) and not (C == 2 and nodal):
                    todo.append((STREAM, False))
                # the dense-tile MFMA solver: float value solves under a
                # label-blind edge kernel
                if C == 1 and not (ngrad or maximin) and \
                        backend._label_blind(kedge):
                    todo.append((MFMA, False))
                for v, tab in todo:
                    src = backend.render_source(
                        knode, ek, k.p, dgraphs[0].node_t, dgraphs[0].edge_t,
                        [v], C, nodal and v not in (GENERAL, MFMA, STREAM),
                        tab=tab,
                        weighted=dgraphs[0].weighted,
                        ngrad=ngrad and isinstance(v, OCVariant),
                        maximin=maximin and isinstance(v, OCVariant))
                    yield (f'{fname}/{rname}/{mname}/'
                           + backend.kernel_name(v, C, nodal, tab, ngrad,
                                                 maximin),
                           tuple(backend.hipcc_extra), src)
                if gtab:
                    from graphdot_amd.kernel.marginalized._backend_hip import \
                        TABLES
                    src = backend.render_source(
                        knode, ek, k.p, dgraphs[0].node_t, dgraphs[0].edge_t,
                        [TABLES], C, False, tab=True,
                        weighted=dgraphs[0].weighted)
                    yield (f'{fname}/{rname}/{mname}/tables',
                           tuple(backend.hipcc_extra), src)


def compile_all(select=None, max_workers=None):
    """Compile the matrix into the JIT cache; returns (labels, failures)
    with failures = [(label, hipcc diagnostics)]."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from graphdot_amd.hip import jit
    items, seen = [], set()
    for label, flags, src in sources(select):
        key = jit.cache_key(src, flags)
        if key not in seen:
            seen.add(key)
            items.append((label, flags, src))

    def one(item):
        label, flags, src = item
        try:
            jit.compile_source(src, flags, keep_source=False)
            return None
        except jit.CompileError as e:
            return (label, str(e)[-3000:])
    with ThreadPoolExecutor(max_workers or os.cpu_count() or 1) as ex:
        failures = [f for f in ex.map(one, items) if f is not None]
    return [i[0] for i in items], failures


if __name__ == '__main__':
    import os
    import sys
    import time
    _here = os.path.dirname(os.path.abspath(__file__))
    for _p in (_here, os.path.dirname(_here)):
        if _p not in sys.path:
            sys.path.insert(0, _p)
    t0 = time.time()
    labels, failures = compile_all(set(sys.argv[1:]) or None)
    print(f'{len(labels)} translation units, {len(failures)} failures, '
          f'{time.time() - t0:.0f} s')
    for label, err in failures:
        print('FAILED', label)
        print(err[-1500:])
    sys.exit(1 if failures else 0)
