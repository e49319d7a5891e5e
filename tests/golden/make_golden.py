#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the *reference*.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite only read the JSON it wrote.  Nothing of the reference is copied:
the script imports the reference's Python under a scratch shim and records
inputs + outputs.

Shim:
  * numpy >= 1.24 removed np.float / np.int / np.object / np.bool /
    np.issctype / np.issubsctype, which the reference uses -> aliases.
  * pycuda (absent) -> stub modules whose managed_* allocators return host
    arrays; mendeleev / pymatgen / ase (absent) -> empty stubs.
  * scipy.sparse.linalg.cg is wrapped to force rtol=1e-13 so that the
    reference's CPU oracles (which pass atol=1e-7 and inherit scipy's default
    rtol=1e-5) are converged to fp64 accuracy.  The reference code itself is
    executed unmodified.

Fixtures written:
  mlgk_cases.json      reference test oracle `MLGK` (test_kernel.py:20-68) on its
                       four case families x q in {0.01,0.05,0.1,0.5}
  m3_cross.json        reference CPU solver `M3._mlgk` (m3.py:52-106) on cross
                       pairs of the example/nodelabeled-weighted.py graphs and
                       seeded random weighted graphs
  host_model.json      codegen strings, theta structs/states, pack_state,
                       OctileGraph degrees & nonzeros, theta plumbing
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def install_shims():
    for name, val in [('float', float), ('int', int), ('object', object),
                      ('bool', bool)]:
        if not hasattr(np, name):
            setattr(np, name, val)

    def issctype(t):
        try:
            dt = np.dtype(t)
        except TypeError:
            return False
        return dt.kind in 'biufcSUmM' and dt.names is None

    np.issctype = issctype
    np.issubsctype = lambda a, b: np.issubdtype(
        a if isinstance(a, (type, np.dtype)) else np.asarray(a).dtype, b)

    class Managed(np.ndarray):
        """Stand-in for a pycuda managed allocation: int() gives its address
        (the reference reads device pointers as int(array.base))."""
        def __int__(self):
            return self.ctypes.data

    def _managed(arr):
        owner = arr.view(Managed)
        out = owner[...]
        assert isinstance(out.base, Managed)
        return out

    drv = types.ModuleType('pycuda.driver')

    def managed_empty(shape, dtype, order='C', mem_flags=0):
        return _managed(np.empty(shape, dtype, order))

    def managed_zeros(shape, dtype, order='C', mem_flags=0):
        return _managed(np.zeros(shape, dtype, order))

    def managed_empty_like(a, mem_flags=0):
        return _managed(np.empty(a.shape, a.dtype))

    drv.managed_empty = managed_empty
    drv.managed_zeros = managed_zeros
    drv.managed_empty_like = managed_empty_like
    drv.mem_attach_flags = types.SimpleNamespace(GLOBAL=1)
    pycuda = types.ModuleType('pycuda')
    pycuda.driver = drv
    autoinit = types.ModuleType('pycuda.autoinit')
    autoinit.context = types.SimpleNamespace(
        get_device=lambda: None, synchronize=lambda: None)
    compiler = types.ModuleType('pycuda.compiler')
    compiler.SourceModule = object
    gpuarray = types.ModuleType('pycuda.gpuarray')
    gpuarray.empty = lambda n, dtype: types.SimpleNamespace(
        ptr=0, data=np.empty(n, dtype))
    sys.modules.update({
        'pycuda': pycuda, 'pycuda.driver': drv, 'pycuda.autoinit': autoinit,
        'pycuda.compiler': compiler, 'pycuda.gpuarray': gpuarray,
    })
    pycuda.autoinit, pycuda.compiler, pycuda.gpuarray = (
        autoinit, compiler, gpuarray)

    for name in ['mendeleev', 'mendeleev.fetch', 'pymatgen', 'pymatgen.core',
                 'pymatgen.io', 'pymatgen.io.ase', 'ase', 'ase.build',
                 'ase.data', 'ase.neighborlist', 'ase.io']:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules['mendeleev'].get_table = lambda *a, **k: None
    sys.modules['mendeleev.fetch'].fetch_table = lambda *a, **k: None
    sys.modules['mendeleev'].fetch = sys.modules['mendeleev.fetch']
    sys.modules['ase'].Atoms = object
    sys.modules['ase.build'].molecule = lambda *a, **k: None
    sys.modules['pymatgen.io.ase'].AseAtomsAdaptor = object
    sys.modules['pymatgen.io'].ase = sys.modules['pymatgen.io.ase']

    import scipy.sparse.linalg as spla
    import scipy.sparse as sp
    _cg = spla.cg

    def tight_cg(A, b, *args, **kwargs):
        kwargs.pop('atol', None)
        kwargs.pop('tol', None)
        kwargs['rtol'] = 1e-13
        kwargs['atol'] = 0.0
        kwargs.setdefault('maxiter', 100000)
        return _cg(A, b, *args, **kwargs)

    spla.cg = tight_cg
    sp.linalg.cg = tight_cg


def jsonable(o):
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    return o


def graph_to_dict(g):
    """Columns of a reference Graph as plain lists (inputs of a fixture)."""
    def frame(df):
        return {c: [jsonable(v if not isinstance(v, np.ndarray) else v.tolist())
                    for v in df[c]] for c in df.columns}
    return {'title': g.title, 'nodes': frame(g.nodes), 'edges': frame(g.edges)}


def load_test_oracle():
    """Execute the reference test module up to (not including) its first test
    function: gives MLGK() and case_dict."""
    path = os.path.join(REF, 'test/kernel/marginalized/test_kernel.py')
    src = open(path).read()
    head = src.split('def test_mlgk_typecheck')[0]
    head = head.replace('from ase.build import molecule\n', '')
    ns = {'__name__': 'ref_test_kernel'}
    exec(compile(head, path, 'exec'), ns)
    return ns


def main():
    install_shims()
    sys.path.insert(0, REF)
    import networkx as nx
    from graphdot import Graph
    from graphdot.microkernel import (
        Constant, KroneckerDelta, SquareExponential, TensorProduct, Additive,
        Convolution, RationalQuadratic, Normalize, Product, DotProduct)
    from graphdot.codegen.cpptool import decltype
    from graphdot.kernel.marginalized._backend_cuda import CUDABackend
    from graphdot.kernel.marginalized._octilegraph import OctileGraph
    from graphdot.kernel.marginalized._backend import Backend
    from graphdot.kernel.marginalized import MarginalizedGraphKernel
    from graphdot.kernel.marginalized.starting_probability import Uniform
    from graphdot.experimental.metric.m3 import M3
    from graphdot.util.iterable import flatten

    ns = load_test_oracle()
    MLGK, case_dict = ns['MLGK'], ns['case_dict']

    # ---------------------------------------------------------------- MLGK
    cases = {}
    for name, case in case_dict.items():
        entry = {
            'graphs': [graph_to_dict(g) for g in case['graphs']],
            'knode': repr(case['knode']),
            'kedge': repr(case['kedge']),
            'q': case['q'], 'R': [], 'R_nodal': [],
        }
        for q in case['q']:
            entry['R'].append([
                float(MLGK(g, case['knode'], case['kedge'], q, q))
                for g in case['graphs']])
            entry['R_nodal'].append([
                MLGK(g, case['knode'], case['kedge'], q, q, nodal=True)
                for g in case['graphs']])
        cases[name] = entry

    # self-loop / weighted random graphs (test_kernel.py:507-525 family)
    rng = np.random.RandomState(2)
    loops = []
    for _ in range(4):
        n = rng.randint(4, 12)
        A = rng.randn(n, n)
        A = np.abs(A + A.T)             # positive weights, with self loops
        g = Graph.from_networkx(nx.from_numpy_array(A), weight='weight')
        loops.append({
            'graph': graph_to_dict(g), 'q': 0.1,
            'R': float(MLGK(g, Constant(1.0), Constant(1.0), 0.1, 0.1))})
    cases['self-loops'] = loops

    with open(os.path.join(HERE, 'mlgk_cases.json'), 'w') as f:
        json.dump(jsonable(cases), f)

    # ------------------------------------------------------------- M3._mlgk
    def nlw_graphs():
        g1 = nx.Graph()
        g1.add_node(0, radius=1.0, category=1)
        g1.add_node(1, radius=2.0, category=1)
        g1.add_edge(0, 1, w=1.0)
        g2 = nx.Graph()
        g2.add_node(0, radius=1.0, category=1)
        g2.add_node(1, radius=2.0, category=1)
        g2.add_node(2, radius=1.0, category=2)
        g2.add_edge(0, 1, w=1.0)
        g2.add_edge(1, 2, w=2.0)
        g3 = nx.Graph()
        g3.add_node(0, radius=1.0, category=1)
        g3.add_node(1, radius=2.0, category=1)
        g3.add_node(2, radius=1.0, category=2)
        g3.add_edge(0, 1, w=1.0)
        g3.add_edge(0, 2, w=0.5)
        g3.add_edge(1, 2, w=2.0)
        return [Graph.from_networkx(g, weight='w') for g in (g1, g2, g3)]

    def m3_solver(knode, kedge, q):
        m = M3.__new__(M3)
        m.q, m.node_kernel, m.edge_kernel = q, knode, kedge
        return m

    m3 = {}
    G = nlw_graphs()
    knode = TensorProduct(radius=SquareExponential(0.5),
                          category=KroneckerDelta(0.5))
    kedge = Constant(1.0)
    solver = m3_solver(knode, kedge, 0.05)
    m3['nodelabeled-weighted'] = {
        'graphs': [graph_to_dict(g) for g in G],
        'knode': repr(knode), 'kedge': repr(kedge), 'q': 0.05,
        'R_nodal': [[solver._mlgk(a, b) for b in G] for a in G],
    }

    rng = np.random.RandomState(7)
    H = []
    for n in (5, 9, 12, 17):
        g = nx.newman_watts_strogatz_graph(n, 3, 0.3, seed=int(rng.randint(1 << 30)))
        for i in g.nodes:
            g.nodes[i]['radius'] = float(rng.choice([1.0, 1.5, 2.0]))
            g.nodes[i]['category'] = int(rng.choice([1, 2, 3]))
        for e in g.edges:
            g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0]))
            g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
        H.append(Graph.from_networkx(g, weight='w'))
    H = Graph.unify_datatype(H)
    knode = TensorProduct(radius=SquareExponential(0.5),
                          category=KroneckerDelta(0.5))
    kedge = TensorProduct(length=SquareExponential(1.0))
    solver = m3_solver(knode, kedge, 0.01)
    m3['random-weighted'] = {
        'graphs': [graph_to_dict(g) for g in H],
        'knode': repr(knode), 'kedge': repr(kedge), 'q': 0.01,
        'R_nodal': [[solver._mlgk(a, b) for b in H] for a in H],
    }
    with open(os.path.join(HERE, 'm3_cross.json'), 'w') as f:
        json.dump(jsonable(m3), f)

    # -------------------------------------------------------------- host model
    host = {}
    kernels = {
        'constant': Constant(1.0),
        'kdelta': KroneckerDelta(0.5),
        'sqexp': SquareExponential(0.5),
        'rq': RationalQuadratic(1.0, 2.0),
        'tp': TensorProduct(radius=SquareExponential(0.5),
                            category=KroneckerDelta(0.5)),
        'additive_norm': Additive(order=KroneckerDelta(0.3),
                                  length=SquareExponential(0.05)).normalized,
        'tp_norm': TensorProduct(hybridization=KroneckerDelta(0.3),
                                 charge=SquareExponential(1.) + 0.01).normalized,
        'conv': TensorProduct(rings=Convolution(KroneckerDelta(0.3))),
        'weighted_wrap': TensorProduct(weight=Product(),
                                       label=TensorProduct(
                                           length=SquareExponential(1.0))),
        'expr': KroneckerDelta(0.5) * 2 + 1,
        'pow': KroneckerDelta(0.5)**2,
        'dot': TensorProduct(v=DotProduct()),
    }
    host['kernels'] = {}
    for key, k in kernels.items():
        f, j = k.gen_expr('x1', 'x2')
        host['kernels'][key] = {
            'repr': repr(k), 'expr': f, 'jac': j, 'decltype': decltype(k),
            'state': jsonable(k.state), 'theta': list(flatten(k.theta)),
            'bounds': [str(b) if isinstance(b, str) else list(b)
                       for b in _flat_bounds(k.bounds)],
            'minmax': _minmax(k),
            'itemsize': k.dtype.itemsize,
        }
    host['pack_state'] = jsonable(CUDABackend.pack_state(
        KroneckerDelta(0.5), diff_grid=True, diff_eps=1e-2))
    host['gencode_p'] = list(Uniform(1.0).gen_expr())

    og = {}
    for name, g in (('nlw3', G[2]), ('rand12', H[2])):
        o = OctileGraph(g)
        og[name] = {
            'graph': graph_to_dict(g),
            'degree': np.asarray(o.degree), 'n_octile': int(o.n_octile),
            'weighted': bool(o.weighted),
            'node_t': str(o.node_t), 'edge_t': str(o.edge_t),
            'node_t_decl': decltype(o.node_t), 'edge_t_decl': decltype(o.edge_t),
            'nzmask': [int(x) for x in o.octiles['nzmask']],
            'nzmask_r': [int(x) for x in o.octiles['nzmask_r']],
            'upper': [int(x) for x in o.octiles['upper']],
            'left': [int(x) for x in o.octiles['left']],
            'weights': np.asarray(o.edges_aos['weight']),
        }
    host['octilegraph'] = og

    class Null(Backend):
        def __call__(self, *a):
            pass

    mk = MarginalizedGraphKernel(
        TensorProduct(f=KroneckerDelta(0.5)),
        TensorProduct(a=SquareExponential(1.0, length_scale_bounds='fixed'),
                      b=KroneckerDelta(0.25)),
        q=0.05, backend=Null())
    host['theta_plumbing'] = {
        'flat': mk.flat_hyperparameters, 'n_dims': mk.n_dims,
        'mask': mk.active_theta_mask, 'theta': mk.theta, 'bounds': mk.bounds,
    }
    with open(os.path.join(HERE, 'host_model.json'), 'w') as f:
        json.dump(jsonable(host), f)
    print('golden fixtures written to', HERE)


def _minmax(k):
    try:
        return [None if v is None else float(v) for v in k.minmax]
    except TypeError:      # e.g. Product() has no range: reference raises
        return None


def _flat_bounds(b):
    out = []
    for item in b:
        if isinstance(item, str):
            out.append(item)
        elif (isinstance(item, tuple) and len(item) == 2
              and all(np.isscalar(v) for v in item)):
            out.append(item)
        else:
            out.extend(_flat_bounds(item))
    return out


if __name__ == '__main__':
    main()
