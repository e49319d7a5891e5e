"""Host-side object model against fixtures recorded from the reference
(codegen grammar, theta structs, states, hyperparameter plumbing, degrees)."""
import os
import re
import numpy as np
import pytest
from numpy import inf  # noqa: F401
from _fixtures import load, graph_from_dict, graphs_from
from graphdot_amd.codegen import Template
from graphdot_amd.codegen.cpptool import cpptype, decltype
from graphdot_amd.graph import Graph
from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel, Backend
from graphdot_amd.kernel.marginalized._devicegraph import (
    DeviceGraph, GraphArena, HEADER_DTYPE)
from graphdot_amd.kernel.marginalized.starting_probability import Uniform
from graphdot_amd.microkernel import (  # noqa: F401
    Constant, KroneckerDelta, SquareExponential, RationalQuadratic,
    TensorProduct, Additive, Composite, Convolution, Normalize, Product,
    DotProduct)
from graphdot_amd.util.iterable import flatten, fold_like

HOST = load('host_model.json')


def squash(text):
    return re.sub(r'\s+', '', text)


def build(key):
    return {
        'constant': lambda: Constant(1.0),
        'kdelta': lambda: KroneckerDelta(0.5),
        'sqexp': lambda: SquareExponential(0.5),
        'rq': lambda: RationalQuadratic(1.0, 2.0),
        'tp': lambda: TensorProduct(radius=SquareExponential(0.5),
                                    category=KroneckerDelta(0.5)),
        'additive_norm': lambda: Additive(
            order=KroneckerDelta(0.3),
            length=SquareExponential(0.05)).normalized,
        'tp_norm': lambda: TensorProduct(
            hybridization=KroneckerDelta(0.3),
            charge=SquareExponential(1.) + 0.01).normalized,
        'conv': lambda: TensorProduct(rings=Convolution(KroneckerDelta(0.3))),
        'weighted_wrap': lambda: TensorProduct(
            weight=Product(),
            label=TensorProduct(length=SquareExponential(1.0))),
        'expr': lambda: KroneckerDelta(0.5) * 2 + 1,
        'pow': lambda: KroneckerDelta(0.5)**2,
        'dot': lambda: TensorProduct(v=DotProduct()),
    }[key]()


@pytest.mark.parametrize('key', sorted(HOST['kernels']))
def test_microkernel_codegen_matches_reference(key):
    ref = HOST['kernels'][key]
    k = build(key)
    expr, jac = k.gen_expr('x1', 'x2')
    assert squash(expr) == squash(ref['expr'])
    assert [squash(j) for j in jac] == [squash(j) for j in ref['jac']]
    assert decltype(k) == ref['decltype']
    assert repr(k) == ref['repr']
    assert k.dtype.itemsize == ref['itemsize']
    assert np.allclose(list(flatten(k.theta)), ref['theta'])
    assert np.allclose(list(flatten(k.state)), list(flatten(ref['state'])))
    if ref['minmax'] is not None:
        assert [float(v) for v in k.minmax] == pytest.approx(ref['minmax'])
    k2 = eval(repr(k))
    assert repr(k2) == repr(k)


def test_microkernel_values_and_jacobians():
    k = TensorProduct(radius=SquareExponential(0.5),
                      category=KroneckerDelta(0.5))
    X, Y = dict(radius=1.0, category=1), dict(radius=1.5, category=2)
    f, j = k(X, Y, jac=True)
    assert f == pytest.approx(np.exp(-0.5) * 0.5)
    h = 1e-6
    for i in range(2):
        t = np.array(list(flatten(k.theta)))
        kp, km = eval(repr(k)), eval(repr(k))
        tp, tm = t.copy(), t.copy()
        tp[i] += h
        tm[i] -= h
        kp.theta = fold_like(tp, kp.theta)
        km.theta = fold_like(tm, km.theta)
        assert j[i] == pytest.approx((kp(X, Y) - km(X, Y)) / (2 * h),
                                     rel=1e-6)
    kn = k.normalized
    assert kn(X, X) == pytest.approx(1.0)
    assert kn.minmax[1] == 1
    assert k(X, Y) == k(Y, X)


def test_pack_state_diff_grid():
    from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend
    got = HIPBackend.pack_state(KroneckerDelta(0.5), diff_grid=True,
                                diff_eps=1e-2)
    assert np.allclose([list(flatten(s)) for s in got],
                       [list(flatten(s)) for s in HOST['pack_state']])
    assert list(Uniform(1.0).gen_expr()) == HOST['gencode_p']


def test_theta_plumbing_matches_reference():
    class Null(Backend):
        def __call__(self, *a):
            pass
    mk = MarginalizedGraphKernel(
        TensorProduct(f=KroneckerDelta(0.5)),
        TensorProduct(a=SquareExponential(1.0, length_scale_bounds='fixed'),
                      b=KroneckerDelta(0.25)),
        q=0.05, backend=Null())
    ref = HOST['theta_plumbing']
    assert np.allclose(mk.flat_hyperparameters, ref['flat'])
    assert mk.n_dims == ref['n_dims']
    assert mk.active_theta_mask.tolist() == ref['mask']
    assert np.allclose(mk.theta, ref['theta'])
    assert np.allclose(mk.bounds, ref['bounds'])
    mk.theta = np.log([2.0, 0.1, 0.25, 0.5])
    assert np.allclose(mk.flat_hyperparameters, [2.0, 0.1, 0.25, 1.0, 0.5])
    clone = mk.clone_with_theta(np.log([1.0, 0.2, 0.3, 0.4]))
    assert np.allclose(clone.flat_hyperparameters, [1.0, 0.2, 0.3, 1.0, 0.4])
    assert np.allclose(mk.flat_hyperparameters, [2.0, 0.1, 0.25, 1.0, 0.5])
    assert 'stopping_probability' in repr(mk.hyperparameters)


def test_range_check_warnings():
    """test_kernel.py:572-605"""
    class Null(Backend):
        def __call__(self, *a):
            pass
    ok = TensorProduct(attribute=SquareExponential(1.0))
    MarginalizedGraphKernel(KroneckerDelta(1e-7), ok, backend=Null())
    for node, edge in [
            (KroneckerDelta(0), ok),
            (TensorProduct(feature=KroneckerDelta(0.5)) + 1,
             SquareExponential(1.0)),
            (TensorProduct(feature=KroneckerDelta(0.5)), ok + 1),
            (KroneckerDelta(0.5) * 2, ok),
            (TensorProduct(feature=KroneckerDelta(0.5)), ok * 2)]:
        with pytest.warns(DeprecationWarning):
            MarginalizedGraphKernel(node, edge, backend=Null())
    with pytest.raises(ValueError):
        MarginalizedGraphKernel(KroneckerDelta(0.5), ok, p=-1.0,
                                backend=Null())
    with pytest.raises(ValueError):
        MarginalizedGraphKernel(KroneckerDelta(0.5), ok, backend='nope')


@pytest.mark.parametrize('name', ['nlw3', 'rand12'])
def test_device_graph_vs_reference_octilegraph(name):
    """Same degrees, node/edge struct declarations and nonzero weights as the
    reference's OctileGraph (layout differs by design, see graph.h)."""
    ref = HOST['octilegraph'][name]
    g = graph_from_dict(ref['graph'])
    d = DeviceGraph(g)
    assert d.weighted == ref['weighted']
    assert decltype(d.node_t) == ref['node_t_decl']
    assert decltype(d.edge_t) == ref['edge_t_decl']
    # degrees are stored in the degree-sorted numbering: map back
    deg = np.empty(d.n_node, np.float32)
    deg[d.perm] = d.degree
    assert deg.tolist() == ref['degree']
    assert d.n_nz == len(ref['weights'])
    edges = d.blob[d.offsets['edge']:d.offsets['edge']
                   + d.n_nz * d.edge_t.itemsize].view(d.edge_t)
    assert sorted(edges['weight'].tolist()) == sorted(ref['weights'])
    # CSR consistency in the new numbering
    assert d.rowptr[-1] == d.n_nz
    assert np.all(np.diff(d.adjacency_count) <= 0)
    A = np.zeros((d.n_node, d.n_node))
    A[d.perm[d.nz['i']], d.perm[d.nz['j']]] = edges['weight']
    assert np.allclose(A, g.adjacency_matrix.toarray())


def test_arena_relocation_of_frozen_arrays():
    G = graphs_from(load('mlgk_cases.json')['vario-features']['graphs'])
    dgs = [DeviceGraph(g) for g in G]
    arena = GraphArena(dgs)
    base = 0x7f0000000000
    img = arena.relocated(base)
    hdr = img[:arena.n * HEADER_DTYPE.itemsize].view(HEADER_DTYPE)
    assert hdr['n_node'].tolist() == [3, 2]
    for k, d in enumerate(dgs):
        nodes = img[hdr['node'][k]:hdr['node'][k]
                    + d.n_node * d.node_t.itemsize].view(d.node_t)
        fa = nodes[d.node_t.names[0]]
        for row in range(d.n_node):
            off = int(fa['ptr'][row]) - base
            size = int(fa['size'][row])
            payload = img[off:off + 2 * size].view(np.int16)
            orig = d.perm[row]
            rings = [r for i, r in zip(G[k].nodes['!i'], G[k].nodes['rings'])
                     if i == orig][0]
            assert payload.tolist() == list(rings)


def test_graph_container_semantics():
    import copy
    import pickle
    import networkx as nx
    g = nx.Graph(title='t')
    g.add_node('a', x=1)
    g.add_node('b', x=300)
    g.add_edge('a', 'b', w=0.5, y=1.5)
    G = Graph.from_networkx(g, weight='w')
    assert G.title == 't'
    assert G.nodes['x'].dtype == np.int32          # uint16 -> signed int32
    assert G.edges['!w'].dtype == np.float32
    assert Graph.has_unified_types([G, G]) is True
    G.cookie['k'] = 1
    assert copy.deepcopy(G).cookie == {}
    assert pickle.loads(pickle.dumps(G)).cookie == {}
    P = G.permute([1, 0])
    assert P.nodes['!i'].tolist() == [1, 0]
    assert np.allclose(G.laplacian.toarray(), [[0.5, -0.5], [-0.5, 0.5]])
    h = nx.Graph()
    h.add_node(0, x=1.5)
    h.add_node(1, x=2.5)
    h.add_edge(0, 1, w=1.0, y=2.0)
    H = Graph.from_networkx(h, weight='w')
    assert Graph.has_unified_types([G, H]) is not True
    U = Graph.unify_datatype([G, H])
    assert Graph.has_unified_types(U) is True
    with pytest.raises(RuntimeError):
        Graph.from_networkx(nx.empty_graph(3))
    back = G.to_networkx()
    assert back.number_of_edges() == 1


def test_template_and_cpptype():
    assert Template('${a}(${b, })').render(a='f', b=[1, 2]) == 'f(1, 2)'
    with Template('?{t.x is True}|?{t.x == 2}').context(
            t=type('T', (), {'x': True})) as t:
        assert t.render() == 'true|false'

    @cpptype(h=np.float32, n=np.int32)
    class K:
        def __init__(self):
            self.h, self.n = 0.5, 3
    k = K()
    assert k.state == (np.float32(0.5), np.int32(3))
    assert decltype(k) == 'struct{float32 h;int32 n;}'
    with pytest.raises(TypeError):
        k.h = 'x'


def test_job_lists_are_cached_read_only_and_match_the_reference_order():
    """_kernel.py:172-182 of the reference: upper triangle incl. diagonal,
    row-major, for a symmetric matrix; (i, nx + j) for X x Y."""
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend import Backend
    from graphdot_amd.microkernel import Constant

    class Dummy(Backend):
        array = staticmethod(lambda a: a)

        def __call__(self, *args, **kwargs):
            raise AssertionError

    k = MarginalizedGraphKernel(Constant(1.0), Constant(1.0), q=0.05,
                                backend=Dummy())
    jobs = k._pairwise_jobs(4)
    assert [tuple(x) for x in jobs] == [
        (0, 0), (0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (1, 3), (2, 2),
        (2, 3), (3, 3)]
    assert not jobs.flags.writeable
    assert k._pairwise_jobs(4) is jobs
    full = k._pairwise_jobs(2, 3)
    assert [tuple(x) for x in full] == [
        (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4)]


def test_row_type_cache_follows_graph_mutation():
    import networkx as nx
    from graphdot_amd.graph import Graph
    a = nx.path_graph(3)
    b = nx.path_graph(4)
    for g, v in ((a, 1), (b, 1.5)):
        for n in g.nodes:
            g.nodes[n]['x'] = v
        for e in g.edges:
            g.edges[e]['w'] = 1.0
    ga, gb = Graph.from_networkx(a, weight='w'), Graph.from_networkx(b, weight='w')
    assert Graph.has_unified_types([ga, ga]) is True
    bad = Graph.has_unified_types([ga, gb])
    assert bad is not True and bad[0] == 'nodes'
    ua, ub = Graph.unify_datatype([ga, gb])
    assert Graph.has_unified_types([ua, ub]) is True
    Graph.unify_datatype([ga, gb], inplace=True)      # clears the cookies
    assert Graph.has_unified_types([ga, gb]) is True


def test_native_table_check_agrees_with_the_row_type_comparison():
    """`Graph.has_unified_types` on a long list goes through one native pass
    (hostlib.same_tables, csrc/gdcollect.cpp); it may only say "unified" where
    the per-graph row-type comparison does, and must leave every other case
    -- another element type, another column order, a missing column, an
    object column -- to it."""
    import cases
    from graphdot_amd.hip import hostlib
    if not hostlib.collector():
        pytest.skip('no CPython extension on this machine')
    G = cases.config3_graphs(40)
    for g in G:
        g.cookie.clear()
    assert hostlib.same_tables(G) is True
    assert Graph.has_unified_types(G) is True
    assert all('rowtypes' not in g.cookie for g in G)     # (the native path)

    def python_says(H):
        for g in H:
            g.cookie.clear()
        first = (H[0].nodes.rowtype(), H[0].edges.rowtype())
        return all((g.nodes.rowtype(), g.edges.rowtype()) == first for g in H)

    retyped = G[5].copy(deep=True)
    retyped.nodes['hcount'] = np.asarray(retyped.nodes['hcount']).astype(np.float64)
    reordered = G[7].copy(deep=True)
    reordered.edges._data = dict(reversed(list(reordered.edges._data.items())))
    dropped = G[9].copy(deep=True)
    del dropped.edges._data['stereo']
    boxed = G[11].copy(deep=True)
    boxed.nodes['tag'] = [(1, 2)] * len(boxed.nodes)
    for k, h in ((5, retyped), (7, reordered), (9, dropped), (11, boxed)):
        H = list(G)
        H[k] = h
        assert hostlib.same_tables(H) is None
        assert python_says(H) is False
        bad = Graph.has_unified_types(H)
        assert bad is not True and bad[2] is h
    assert python_says(G) is True


def test_graph_list_identity_cache_dies_with_the_cookie_epoch():
    """`has_unified_types` and the backend remember what they derived from a
    list of graphs by the identities of its members
    (util.cookie.IdentityCache); any graph dropping cached state (permute,
    re-typing, a deleted packing) invalidates every entry."""
    from graphdot_amd.util.cookie import IdentityCache, VolatileCookie
    import graphdot_amd.graph as graph_module
    G = graphs_from(load('mlgk_cases.json')['labeled']['graphs'])
    cache = IdentityCache(maxsize=2)
    key, hit = cache.get(G)
    assert hit is None
    cache.put(key, G, 'derived')
    assert cache.get(G)[1] == 'derived'
    assert cache.get(list(G))[1] == 'derived'         # same members
    assert cache.get(G[::-1])[1] is None              # other order
    G[0].cookie['x'] = 1                              # adding keeps entries
    assert cache.get(G)[1] == 'derived'
    before = VolatileCookie.epoch
    del G[0].cookie['x']
    assert VolatileCookie.epoch == before + 1 and cache.get(G)[1] is None
    # the type check: cached positive, re-run after a mutation
    assert Graph.has_unified_types(G) is True
    assert graph_module._UNIFIED.get(G)[1] is True
    G[1].permute(np.arange(len(G[1].nodes))[::-1], inplace=True)
    assert graph_module._UNIFIED.get(G)[1] is None
    assert Graph.has_unified_types(G) is True


def test_batch_packer_is_byte_identical_to_the_per_graph_packer():
    """`pack_many` (all graphs of a call in one pass -- natively in
    libgdhost.so, `gdh_pack_graphs`, or the numpy restatement; SURVEY 8f
    rank 1; reference: _octilegraph.py:37-177 once per graph) produces the
    very blobs, permutations and CSR arrays of `DeviceGraph(graph)`, for
    labeled / weighted / unlabeled graphs, self loops, both arithmetics, and
    falls back to the per-graph packer for variable-length attributes; the
    arenas built from them (headers, label classes) are equal byte for
    byte."""
    import cases
    from _fixtures import load, graphs_from
    from graphdot_amd.kernel.marginalized._devicegraph import (
        DeviceGraph, GraphArena, pack_many)
    sets = [cases.config3_graphs(60), cases.config2_graphs(12, seed=3),
            cases.config1_graphs(), cases.nlw_example_graphs()]
    M = load('mlgk_cases.json')
    sets += [graphs_from(M[name]['graphs']) for name in
             ('unlabeled', 'labeled', 'weighted', 'vario-features')]
    # self loops and repeated edges (duplicates collapse onto their first
    # occurrence), an isolated node (degree 0 -> 1)
    from graphdot_amd.graph import Graph
    odd = [Graph(nodes={'!i': [0, 1, 2, 3], 'f': [1.0, 2.0, 3.0, 4.0]},
                 edges={'!i': [0, 1, 1, 0, 2], '!j': [1, 0, 1, 1, 2],
                        '!w': [0.5, 2.0, 1.5, 4.0, 0.25],
                        'len': [1.0, 2.0, 3.0, 4.0, 5.0]}),
           Graph(nodes={'!i': [1, 0, 2], 'f': [1.0, 2.0, 3.0]},
                 edges={'!i': [2, 0], '!j': [0, 1], '!w': [1.0, 3.0],
                        'len': [0.5, 0.25]})]
    sets.append(Graph.unify_datatype(odd))
    for G in sets:
        for real in (np.float32, np.float64):
          for native in (True, False):      # libgdhost.so / numpy restatement
            a = pack_many(G, real, native=native)
            b = [DeviceGraph(g, real) for g in G]
            for x, y in zip(a, b):
                assert x.signature == y.signature
                assert x.offsets == y.offsets
                assert (x.n_node, x.n_nz, x.image_bytes, x.weighted) == \
                    (y.n_node, y.n_nz, y.image_bytes, y.weighted)
                assert np.array_equal(x.blob, y.blob)
                assert np.array_equal(x.perm, y.perm)
                assert np.array_equal(x.rank, y.rank)
                assert np.array_equal(x.adjacency_count, y.adjacency_count)
                assert np.array_equal(x.edge_index, y.edge_index)
                assert np.array_equal(x.relocs, y.relocs)
                assert np.array_equal(x.degree, y.degree)
                assert np.array_equal(x.rowptr, y.rowptr)
                assert np.array_equal(x.nz, y.nz)
                assert x.max_degree == y.max_degree
            # label classes numbered natively (gdh_number_records) and by
            # numpy: the same arena, byte for byte
            A, B = GraphArena(a, native=native), GraphArena(b, native=False)
            assert np.array_equal(A.relocated(1 << 20), B.relocated(1 << 20))
            assert (A.classes is None) == (B.classes is None)
            assert A.classes == B.classes


def test_native_job_layout_equals_the_numpy_restatement():
    """The per-call host work in libgdhost.so (`gdh_classify_oc`: solver
    variant per pair of graph classes; `gdh_pair_keys` / `gdh_order_jobs`:
    class pair of every job and the launch order by a stable counting sort)
    gives exactly the layout of the numpy implementation: the same variants
    in use, the same launch geometry, the same job order -- for the molecular
    set (static layouts), configuration 2 (dynamic one- and multi-wave
    variants), both arithmetics, value and value + gradient, full and X x Y
    job lists, and for a menu the native classifier does not cover."""
    import cases
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend_hip import (
        HIPBackend, VARIANTS, OC_VARIANTS, GENERAL)
    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])

    def triu(n):
        i, j = np.triu_indices(n)
        return np.column_stack((i, j)).astype(np.uint32).ravel().view(job_t)

    def cross(nx, ny):
        i, j = np.indices((nx, ny))
        return np.column_stack((i.ravel(), j.ravel() + nx)).astype(
            np.uint32).ravel().view(job_t)

    sets = [(cases.config3_graphs(150, seed=4), cases.config3_kernels()),
            (cases.config2_graphs(100, seed=2), cases.config2b_kernels())]
    for G, (kn, ke, q) in sets:
        # (>= 4096 jobs: classified per pair of graph classes; fewer: per job)
        for jobs in (triu(len(G)), cross(50, len(G) - 50)):
            for real in (np.float32, np.float64):
                for C in (1, 2):
                    for menu in (None, VARIANTS + OC_VARIANTS + [GENERAL]):
                        out = []
                        for native in (True, False):
                            kw = {} if menu is None else {'variants': menu}
                            b = HIPBackend(real=real, native=native, **kw)
                            k = MarginalizedGraphKernel(kn, ke, q=q, backend=b)
                            dgs, _, _, fields = b._graphs_and_kernels(
                                G, kn, ke, k.traits(symmetric=True))
                            arena = b._host_arena(dgs, fields)
                            out.append(b._partition(
                                dgs, jobs, C, 0, b._global_tables(arena)))
                        (_, ua, oa, La), (_, ub, ob, Lb) = out
                        assert ua == ub and La == Lb
                        assert np.array_equal(oa, ob)
                        assert sorted(oa.tolist()) == list(range(len(jobs)))


def test_graphs_beyond_the_u16_device_format_are_refused():
    """The device format indexes nodes and directed nonzeros with 16 bits
    (graph.h; DESIGN.md 2, "Limits"): a graph with more than 65 535 nodes, or
    more than 65 535 directed nonzeros, raises ValueError -- from the
    per-graph packer and from the batch packer, natively and in numpy --
    instead of wrapping around.  (The reference has no such limit; its octile
    format indexes with 32 bits.)"""
    from graphdot_amd.graph import Graph
    from graphdot_amd.kernel.marginalized._devicegraph import (
        DeviceGraph, pack_many)
    n = 65536 + 4
    ring = Graph(nodes={'!i': np.arange(n), 'f': np.zeros(n, np.float32)},
                 edges={'!i': np.arange(n - 1), '!j': np.arange(1, n)})
    m = 32768 + 8                               # 2 m > 65 535 nonzeros
    rng = np.random.default_rng(0)
    dense = Graph(nodes={'!i': np.arange(400), 'f': np.zeros(400, np.float32)},
                  edges={'!i': rng.integers(0, 200, m),
                         '!j': rng.integers(200, 400, m)})
    ok = Graph(nodes={'!i': np.arange(5), 'f': np.zeros(5, np.float32)},
               edges={'!i': [0, 1, 2, 3], '!j': [1, 2, 3, 4]})
    for bad in (ring, dense):
        with pytest.raises(ValueError, match='65535'):
            DeviceGraph(bad)
        for native in (True, False):
            with pytest.raises(ValueError, match='65535'):
                pack_many(Graph.unify_datatype([ok, bad]), native=native)
    assert len(pack_many([ok])) == 1


def test_native_packer_on_random_multigraphs():
    """The native packer against the per-graph packer on 200 seeded random
    graphs with everything the format has to survive: self loops, repeated
    edges in both orientations, isolated nodes, permuted node ids, weights
    whose float32 sums depend on the summation order, integer / float / bool
    attributes -- blobs, permutations, CSR arrays and degree histograms byte
    for byte; and the job layout of their pairs natively and in numpy."""
    from graphdot_amd.graph import Graph
    from graphdot_amd.kernel.marginalized._devicegraph import (
        DeviceGraph, GraphArena, pack_many)
    rng = np.random.default_rng(20251003)
    graphs = []
    for _ in range(200):
        n = int(rng.integers(1, 14))
        m = int(rng.integers(1, 3 * n + 2))
        ids = rng.permutation(n)
        graphs.append(Graph(
            nodes={'!i': ids, 'z': rng.integers(1, 9, n).astype(np.int8),
                   'c': rng.normal(size=n).astype(np.float32),
                   'a': rng.integers(0, 2, n).astype(bool)},
            edges={'!i': rng.integers(0, n, m), '!j': rng.integers(0, n, m),
                   '!w': rng.uniform(0.1, 3.0, m).astype(np.float32),
                   'o': rng.choice([1.0, 1.5, 2.0], m).astype(np.float32),
                   's': rng.integers(0, 3, m).astype(np.int8)}))
    graphs = Graph.unify_datatype(graphs)
    for real in (np.float32, np.float64):
        a = pack_many(graphs, real, native=True)
        b = [DeviceGraph(g, real) for g in graphs]
        for x, y in zip(a, b):
            assert np.array_equal(x.blob, y.blob)
            assert np.array_equal(x.perm, y.perm)
            assert np.array_equal(x.rowptr, y.rowptr)
            assert np.array_equal(x.nz, y.nz)
            assert np.array_equal(x.edge_index, y.edge_index)
            assert np.array_equal(x.degree, y.degree)
            assert np.array_equal(x.degree_hist, y.degree_hist)
        A = GraphArena(a, native=True)
        B = GraphArena(b, native=False)
        assert np.array_equal(A.relocated(4096), B.relocated(4096))
        assert A.classes == B.classes


def test_native_pairwise_job_list_matches_numpy():
    """gdh_pairwise_jobs against the numpy statement of the reference's job
    list (_kernel.py:172-182): upper triangle with the diagonal, row-major;
    X against Y with Y's graphs numbered after X's; empty lists."""
    from graphdot_amd.hip import hostlib
    for n in (0, 1, 2, 7, 300):
        i, j = np.triu_indices(n)
        got = hostlib.pairwise_jobs(n).reshape(-1, 2)
        assert np.array_equal(got, np.column_stack((i, j)))
    for nx, ny in ((0, 3), (3, 0), (1, 1), (5, 3), (40, 70)):
        i, j = np.indices((nx, ny))
        got = hostlib.pairwise_jobs(nx, ny).reshape(-1, 2)
        assert np.array_equal(got, np.column_stack((i.ravel(),
                                                    j.ravel() + nx)))
    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
    jobs = hostlib.pairwise_jobs(3, None, job_t)
    assert jobs.dtype == job_t and jobs['j'].tolist() == [0, 1, 2, 1, 2, 2]


def test_native_job_order_is_a_stable_sort_in_both_regimes():
    """gdh_order_jobs against numpy's stable argsort: the counting sort (up to
    65 536 ranks) and the two- and three-pass radix sort behind it; jobs of one
    rank keep their order, the job records travel with the ids."""
    from graphdot_amd.hip import hostlib
    rng = np.random.default_rng(0)
    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
    for n, nk, nr in ((0, 3, 3), (5, 4, 2049), (3000, 50, 7), (20000, 9000, 65536),
                      (20000, 9000, 65537), (30000, 100000, 100000),
                      (30000, 5000000, 4500000)):
        pk = rng.integers(0, nk, n).astype(np.int32)
        rank_of_key = rng.integers(0, nr, nk).astype(np.int32)
        jobs = rng.integers(0, 1000, (n, 2)).astype(np.uint32).ravel().view(job_t)
        order, moved = hostlib.order_jobs(pk, rank_of_key, nr, jobs)
        ref = np.argsort(rank_of_key[pk], kind='stable')
        assert np.array_equal(order, ref.astype(np.uint32)), (n, nk, nr)
        assert np.array_equal(moved, jobs[ref])
        assert np.array_equal(hostlib.order_jobs(pk, rank_of_key, nr), order)


def test_composite_kernels_of_one_shape_share_a_class_not_their_state():
    """Composite microkernels are instances of one class per (operator,
    attribute, state type) shape -- made once, the marginalized kernel wraps
    its edge kernel on every evaluation of weighted graphs -- and stay
    independent objects: hyperparameters, repr, evaluation, generated code and
    packed state follow the instance."""
    import copy
    a = TensorProduct(radius=SquareExponential(0.5), category=KroneckerDelta(0.5))
    b = TensorProduct(radius=SquareExponential(2.0), category=KroneckerDelta(0.25))
    assert type(a) is type(b)
    c = Additive(radius=SquareExponential(0.5), category=KroneckerDelta(0.5))
    d = TensorProduct(category=KroneckerDelta(0.5), radius=SquareExponential(0.5))
    assert type(c) is not type(a) and type(d) is not type(a)
    assert d.dtype.names != a.dtype.names or type(d) is type(a)
    ta = np.array(list(flatten(a.theta)))
    b.theta = fold_like(np.array([3.0, 0.125]), b.theta)
    assert np.allclose(list(flatten(a.theta)), ta)           # untouched
    assert np.allclose(list(flatten(b.theta)), [3.0, 0.125])
    assert a.radius is not b.radius
    assert np.allclose(list(flatten(a.radius.theta)), [0.5])
    x = {'radius': 1.0, 'category': 1}
    y = {'radius': 1.5, 'category': 1}
    assert np.isclose(a(x, y), np.exp(-0.5 * 0.25 / 0.25))
    assert np.isclose(b(x, y), np.exp(-0.5 * 0.25 / 9.0))
    assert np.isclose(c(x, y), np.exp(-0.5) + 1.0)
    assert eval(repr(b)).state == b.state and eval(repr(a)).state == a.state
    assert a.gen_expr('x', 'y')[0] == b.gen_expr('x', 'y')[0]
    assert a.state != b.state
    e = copy.deepcopy(a)
    e.theta = fold_like(np.array([9.0, 0.9]), e.theta)
    assert np.allclose(list(flatten(a.theta)), ta)


@pytest.mark.skipif(not os.path.isdir('/root/reference/graphdot'),
                    reason='needs the reference checkout (build container)')
def test_reference_kernel_object_drives_the_hip_backend():
    """The seam regression test: the REFERENCE's MarginalizedGraphKernel
    (`graphdot/kernel/marginalized/_kernel.py:224-242,363-381`) constructs the
    arguments of `HIPBackend.__call__` from its own Graph / microkernel
    objects, for value, gradient, nodal, X x Y, lmin and the three `diag`
    modes; everything of `prepare` that needs no device runs on them
    (tests/golden/check_dropin.py, in a process of its own: the numpy /
    pycuda shims the reference's import needs must not leak into this one).
    Skipped where /root/reference does not exist (the GPU box)."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                          'golden', 'check_dropin.py')
    r = subprocess.run([sys.executable, script], capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'drop-in seam ok' in r.stdout
    assert r.stdout.count('10 backend calls') == 4


@pytest.mark.parametrize('weighted', [False, True])
def test_native_arena_assembly_equals_the_numpy_restatement(weighted):
    """gdh_gather_section / gdh_number_records (hash numbering) /
    gdh_assemble_arena against the numpy GraphArena: the same image, headers,
    class representatives, byte for byte -- molecules (label classes in use)
    and weighted random graphs with continuous edge labels (many classes)."""
    import cases
    from graphdot_amd.kernel.marginalized._devicegraph import (
        pack_many, GraphArena)
    if weighted:
        G = cases.config2_graphs(40, seed=4)
        fields = (('category',), ('length',))
    else:
        G = cases.config3_graphs(120, seed=8)
        fields = (('aromatic', 'atomic_number', 'hcount'),
                  ('conjugated', 'order'))
    for real in (np.float32, np.float64):
        dgs = pack_many(G, real=real)
        for f in (fields, (None, None), ((), ())):
            a = GraphArena(dgs, *f, native=True)
            b = GraphArena(dgs, *f, native=False)
            assert a.nbytes == b.nbytes and (a.classes is None) == (b.classes is None)
            if a.classes is not None:
                assert a.classes == b.classes
            assert np.array_equal(a.host, b.host)
            assert np.array_equal(a.relocated(4096), b.relocated(4096))
