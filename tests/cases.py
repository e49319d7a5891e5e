"""Workloads shared by the GPU parity tests, ``__graft_entry__.build()``
(which pre-compiles their code objects) and ``bench.py``.

Everything is synthetic and seeded; nothing here reads /root/reference.
"""
import numpy as np
import networkx as nx
from graphdot_amd.graph import Graph
from graphdot_amd.microkernel import (
    Constant, KroneckerDelta, SquareExponential, TensorProduct)


# -- config 1: example/unlabeled-unweighted.py + 10 ER graphs ------
def config1_graphs():
    g1 = nx.Graph(); g1.add_edges_from([(0, 1)])
    g2 = nx.Graph(); g2.add_edges_from([(0, 1), (1, 2)])
    g3 = nx.Graph(); g3.add_edges_from([(0, 1), (0, 2), (1, 2)])
    out = [g1, g2, g3]
    rng = np.random.default_rng(0)
    for _ in range(10):
        n = int(rng.integers(3, 9))
        g = nx.Graph()
        g.add_nodes_from(range(n))
        for i in range(n):
            for j in range(i):
                if rng.random() < 0.5:
                    g.add_edge(i, j)
        for i in range(n):          # re-wire isolated nodes
            if g.degree[i] == 0:
                g.add_edge(i, (i + 1) % n)
        out.append(g)
    return Graph.unify_datatype([Graph.from_networkx(g) for g in out])


def config1_kernels():
    return Constant(1.0), Constant(1.0), 0.05


# -- config 2: node-labeled, weighted random graphs ---------------------------------
def nlw_example_graphs():
    """The three graphs of the reference's example/nodelabeled-weighted.py."""
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
    return Graph.unify_datatype(
        [Graph.from_networkx(g, weight='w') for g in (g1, g2, g3)])


def config2_graphs(n_graphs=256, nmin=8, nmax=48, seed=0):
    """Newman-Watts-Strogatz graphs (k=5, p=0.05) as in the reference's
    benchmark/kernel/marginalized/time_kernel.py:14-29, with node attributes
    radius/category, edge weight w and edge attribute length."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_graphs):
        n = int(rng.integers(nmin, nmax + 1))
        g = nx.newman_watts_strogatz_graph(n, 5, 0.05,
                                           seed=int(rng.integers(1 << 30)))
        for i in g.nodes:
            g.nodes[i]['radius'] = float(rng.choice([1.0, 1.5, 2.0]))
            g.nodes[i]['category'] = int(rng.choice([1, 2, 3]))
        for e in g.edges:
            g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0]))
            g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
        out.append(Graph.from_networkx(g, weight='w'))
    return Graph.unify_datatype(out)


def config2a_kernels():
    """script-faithful: example/nodelabeled-weighted.py:44-52"""
    return (TensorProduct(radius=SquareExponential(0.5),
                          category=KroneckerDelta(0.5)),
            Constant(1.0), 0.05)


def config2b_kernels():
    """BASELINE.json-faithful: KroneckerDelta node x SquareExponential edge"""
    return (TensorProduct(category=KroneckerDelta(0.5)),
            TensorProduct(length=SquareExponential(1.0)), 0.05)


# -- config 3: synthetic QM7-like molecules -------------------------------
def synthetic_energies(graphs, seed=0, noise=0.02):
    """Synthetic atomisation-energy-like targets for the QM7-like molecules
    (configuration 5, SURVEY 8d: "y = synthetic energies"; QM7's own are not
    obtainable): one contribution per atom by element, one per bond by order,
    a small non-additive term per aromatic bond, Gaussian noise."""
    rng = np.random.default_rng(seed)
    per_atom = {1: -0.50, 6: -1.60, 7: -1.15, 8: -0.95, 16: -0.85}
    y = np.empty(len(graphs))
    for k, g in enumerate(graphs):
        e = sum(per_atom.get(int(z), -1.0) for z in g.nodes['atomic_number'])
        order = np.asarray(g.edges['order'], dtype=float)
        e += -0.35 * float(order.sum())
        e += -0.20 * float(np.count_nonzero(np.asarray(g.edges['aromatic'])))
        y[k] = e
    return y + noise * np.abs(y).mean() * rng.normal(size=len(y))


_VALENCE = {6: 4, 7: 3, 8: 2, 16: 2}


def qm7_like_molecule(rng, ring_list=False):
    """A random molecule-like graph in the size range of QM7: up to 7 heavy
    atoms (C, N, O, S) joined as a random tree respecting valence, 0-2 ring
    closures, some double/aromatic bonds, then hydrogens on every free
    valence (at most 23 atoms in total).  Node and edge attributes follow the
    reference's Graph.from_rdkit (graph/_from_rdkit.py:219-243); with
    `ring_list` every atom also carries the variable-length attribute
    `ring_list`: the sorted sizes of the rings it lies on, or (0,)
    (_from_rdkit.py:207-212,228-230)."""
    n_heavy = int(np.clip(round(rng.normal(6.3, 1.0)), 1, 7))
    Z = rng.choice([6, 7, 8, 16], size=n_heavy, p=[0.72, 0.12, 0.14, 0.02])
    Z[0] = 6
    cap = np.array([_VALENCE[int(z)] for z in Z])
    used = np.zeros(n_heavy, dtype=int)
    bonds = {}
    for v in range(1, n_heavy):
        cand = [u for u in range(v) if used[u] < cap[u] - (u == 0 and v < 2)]
        cand = [u for u in cand if used[u] < cap[u]]
        if not cand:                    # saturated so far: open up a carbon
            u = int(rng.integers(0, v))
            Z[u], cap[u] = 6, 4
        else:
            u = int(rng.choice(cand))
        bonds[(u, v)] = 1.0
        used[u] += 1
        used[v] += 1
    for _ in range(int(rng.integers(0, 3))):       # ring closures
        free = [u for u in range(n_heavy) if used[u] < cap[u]]
        if len(free) >= 2:
            u, v = sorted(rng.choice(free, size=2, replace=False).tolist())
            if (u, v) not in bonds:
                bonds[(u, v)] = 1.0
                used[u] += 1
                used[v] += 1
    for (u, v) in list(bonds):                     # double / aromatic bonds
        if used[u] < cap[u] and used[v] < cap[v] and rng.random() < 0.15:
            bonds[(u, v)] = float(rng.choice([1.5, 2.0]))
            used[u] += 1
            used[v] += 1
    free = np.maximum(cap - used, 0)
    h_nodes = []
    for u in rng.permutation(n_heavy):
        for _ in range(int(free[u])):
            if n_heavy + len(h_nodes) < 23:
                h_nodes.append(int(u))
    hcount = np.bincount(np.array(h_nodes, dtype=int), minlength=n_heavy)
    g = nx.Graph()
    for u in range(n_heavy):
        mult = [o for (a, b), o in bonds.items() if u in (a, b)]
        g.add_node(u, atomic_number=int(Z[u]), charge=0, hcount=int(hcount[u]),
                   hybridization=int(4 - min(3, sum(o > 1 for o in mult))),
                   aromatic=bool(any(o == 1.5 for o in mult)), chiral=0)
    for k, u in enumerate(h_nodes):
        h = n_heavy + k
        g.add_node(h, atomic_number=1, charge=0, hcount=0, hybridization=1,
                   aromatic=False, chiral=0)
        g.add_edge(u, h, order=1.0, aromatic=False, conjugated=False,
                   stereo=0, ring_stereo=0.0)
    for (u, v), o in bonds.items():
        g.add_edge(u, v, order=float(o), aromatic=bool(o == 1.5),
                   conjugated=bool(o > 1.0), stereo=0, ring_stereo=0.0)
    if g.number_of_edges() == 0:                   # fully unsaturated atom
        g.add_node(n_heavy, atomic_number=1, charge=0, hcount=0,
                   hybridization=1, aromatic=False, chiral=0)
        g.add_edge(0, n_heavy, order=1.0, aromatic=False, conjugated=False,
                   stereo=0, ring_stereo=0.0)
    if ring_list:
        rings = {v: [] for v in g.nodes}
        for cycle in nx.cycle_basis(g):
            for v in cycle:
                rings[v].append(len(cycle))
        for v in g.nodes:
            g.nodes[v]['ring_list'] = tuple(sorted(rings[v])) or (0,)
    return g


def config3_graphs(n_graphs=1000, seed=7165, ring_list=False):
    """The benchmark set (ring_list=False: scalar attributes only; the same
    molecules with the variable-length `ring_list` attribute otherwise)."""
    rng = np.random.default_rng(seed)
    return Graph.unify_datatype(
        [Graph.from_networkx(qm7_like_molecule(rng, ring_list))
         for _ in range(n_graphs)])


def config3_kernels():
    node = TensorProduct(atomic_number=KroneckerDelta(0.5),
                         hcount=SquareExponential(1.0),
                         aromatic=KroneckerDelta(0.8))
    edge = TensorProduct(order=SquareExponential(0.5),
                         conjugated=KroneckerDelta(0.5))
    return node, edge, 0.01


def config3_fit_kernels():
    """Configuration 5 as a *fit*: the kernels of `config3_kernels` with the
    hyperparameter ranges a user of the reference gives an optimiser -- the
    node kernel must stay in (0, 1] (reference _kernel.py:75-92: with a
    length scale of 0.1 exp(-d^2 / 2 l^2) underflows to 0 for hydrogen counts
    that differ by 3, the system is singular and the derivative's 1 / kv^2,
    marginalized_kernel.h:854,871, is NaN on any backend), so the node length
    scale is bounded to [0.5, 10], the edge one to [0.1, 10] and the Kronecker
    deltas to [0.01, 1]."""
    node = TensorProduct(
        atomic_number=KroneckerDelta(0.5, h_bounds=(1e-2, 1)),
        hcount=SquareExponential(1.0, length_scale_bounds=(0.5, 10.0)),
        aromatic=KroneckerDelta(0.8, h_bounds=(1e-2, 1)))
    edge = TensorProduct(
        order=SquareExponential(0.5, length_scale_bounds=(0.1, 10.0)),
        conjugated=KroneckerDelta(0.5, h_bounds=(1e-2, 1)))
    return node, edge, 0.01


# -- the reference's own benchmark shapes ------------------------------------------
def nws48_graphs(batch, size=48):
    """`make_graphs(batch, 48)` of the reference's benchmark harness
    (benchmark/kernel/marginalized/time_kernel.py:14-29): `batch` copies of
    ONE Newman-Watts-Strogatz topology (k = 5, p = 0.05, seed 0) with random
    integer node labels 0..8, edge labels 0..8 and edge weights 1..4."""
    rng = np.random.RandomState(0)
    out = []
    for _ in range(batch):
        g = nx.newman_watts_strogatz_graph(size, k=5, p=0.05, seed=0)
        for i in range(size):
            g.nodes[i]['label'] = int(rng.randint(0, 9))
        for ij in g.edges:
            g.edges[ij]['label'] = int(rng.randint(0, 9))
            g.edges[ij]['weight'] = int(rng.randint(1, 5))
        out.append(Graph.from_networkx(g, weight='weight'))
    return Graph.unify_datatype(out)


def nws48_kernels():
    """time_kernel.py:48-49,62-63; q is the kernel's default."""
    return (TensorProduct(label=KroneckerDelta(0.5)),
            TensorProduct(label=KroneckerDelta(0.5)), 0.01)


# -- dense, from_ase-like molecular graphs (Tang2019MolecularKernel) --------------------
#: van der Waals radii in Angstrom (the `vdw_radius` column the reference's
#: AtomicAdjacency reads from mendeleev, graph/adjacency/atomic.py:32-37)
_VDW = {1: 1.10, 6: 1.70, 7: 1.55, 8: 1.52, 16: 1.80}


def tang2019_graphs(n_graphs=256, seed=2019):
    """Spatial molecular graphs as `Graph.from_ase` builds them
    (graph/_from_ase.py:33-77 with the default AtomicAdjacency,
    graph/adjacency/atomic.py:80-125, euclidean.py:19-31): nodes carry
    `element`, every pair of atoms closer than 3 sqrt(r_i r_j) (van der Waals
    radii) is an edge of weight 1 - d / (3 sqrt(r_i r_j)) with the attribute
    `length` = d.  The molecules are the QM7-like ones above, embedded in 3D
    by a random self-avoiding growth (bond lengths 1.09 / 1.45 A): at most 23
    atoms, 10-20 neighbours per atom -- the near-complete weighted adjacency
    of the reference's flagship molecular kernel (kernel/molecular.py:47-66)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_graphs):
        mol = qm7_like_molecule(rng)
        n = mol.number_of_nodes()
        Z = np.array([mol.nodes[v]['atomic_number'] for v in range(n)])
        x = np.zeros((n, 3))
        placed = {0}
        order = list(nx.bfs_edges(mol, 0))
        for u, v in order:
            bond = 1.09 if 1 in (Z[u], Z[v]) else 1.45
            for attempt in range(200):
                d = rng.normal(size=3)
                cand = x[u] + bond * d / np.linalg.norm(d)
                others = np.array([x[w] for w in placed if w != u])
                if len(others) == 0 or np.min(np.linalg.norm(
                        others - cand, axis=1)) > (0.9 if attempt < 150
                                                   else 0.5):
                    break
            x[v] = cand
            placed.add(v)
        g = nx.Graph()
        for v in range(n):
            g.add_node(v, element=int(Z[v]))
        for i in range(n):
            for j in range(i + 1, n):
                d = float(np.linalg.norm(x[i] - x[j]))
                cut = 3.0 * np.sqrt(_VDW[int(Z[i])] * _VDW[int(Z[j])])
                w = 1.0 - d / cut
                if w > 0:
                    g.add_edge(i, j, w=float(np.float32(w)),
                               length=float(np.float32(d)))
        out.append(Graph.from_networkx(g, weight='w'))
    return Graph.unify_datatype(out)


def tang2019_kernels():
    """Tang2019MolecularKernel's defaults (kernel/molecular.py:36-56)."""
    return (TensorProduct(element=KroneckerDelta(0.2)),
            TensorProduct(length=SquareExponential(0.05)), 0.01)


def feature_graphs(seed=12, n_graphs=5, real=np.float32):
    """Small weighted graphs whose nodes carry a scalar `radius`, a category
    and a fixed-length non-negative feature vector `fp` (variable-length
    attribute on the device), edges a `length`.  With real = float64 the float
    attributes are stored as float64 columns, so that the Python microkernels
    of the oracle see the numbers the double build computes on (a float32
    column makes numpy evaluate `x - y` in float32)."""
    import networkx as nx
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_graphs):
        n = int(rng.integers(5, 12))
        g = nx.connected_watts_strogatz_graph(n, 3, 0.3,
                                              seed=int(rng.integers(1 << 30)))
        for v in g.nodes:
            g.nodes[v]['radius'] = float(rng.choice([1.0, 1.5, 2.0, 2.5]))
            g.nodes[v]['category'] = int(rng.integers(1, 4))
            g.nodes[v]['fp'] = np.round(rng.uniform(0.2, 1.0, size=4),
                                        3).astype(real)
        for e in g.edges:
            g.edges[e]['w'] = float(rng.choice([0.5, 1.0, 2.0]))
            g.edges[e]['length'] = float(rng.uniform(0.5, 2.5))
        out.append(Graph.from_networkx(g, weight='w'))
    if real is np.float64:
        for g in out:
            g.nodes['radius'] = np.asarray(g.nodes['radius'], dtype=real)
            g.edges['length'] = np.asarray(g.edges['length'], dtype=real)
            g.edges['!w'] = np.asarray(g.edges['!w'], dtype=real)
    out = Graph.unify_datatype(out)
    if real is np.float64:
        # (unify_datatype stores list-like attributes with the smallest
        # element type that holds the values: float32)
        for g in out:
            g.nodes['fp'] = [np.asarray(a, dtype=real) for a in g.nodes['fp']]
    return out


def protein_like_graphs(n_graphs=32, nmin=150, nmax=600, seed=3000,
                        cutoff=2.7):
    """Large spatial graphs in the regime of the reference's protein
    benchmark (example/perfbench/protein-time-to-solution.py:1-58: 3 kDa
    crystal structures through `Graph.from_ase`): `n_graphs` point clouds of
    nmin..nmax atoms (H, C, N, O, S in protein-like proportions) grown as a
    compact self-avoiding chain -- consecutive atoms 1.1-1.5 A apart, the
    chain pulled back towards the centre beyond the radius of a globule of
    protein density -- numbered in chain order like a structure file; every
    pair of atoms closer than `cutoff` A is an edge of weight 1 - d / cutoff
    (the tent function of graph/adjacency/atomic.py:80-125) with the
    attribute `length` = d: 6-25 neighbours per atom (mean 13-17), 1400-4000
    edges per graph, product graphs of 2e4-3.6e5 rows with up to 6e7 terms.  Far beyond the register- and
    LDS-resident solvers: the general solver's workload."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_graphs):
        n = int(rng.integers(nmin, nmax + 1))
        Z = rng.choice([1, 6, 7, 8, 16], size=n,
                       p=[0.50, 0.32, 0.085, 0.09, 0.005])
        R = (n / (4.0 / 3.0 * np.pi * 0.085)) ** (1.0 / 3.0)
        x = np.zeros((n, 3))
        for v in range(1, n):
            step = 1.1 if 1 in (Z[v - 1], Z[v]) else 1.5
            best, best_d = None, -1.0
            for attempt in range(30):
                d = rng.normal(size=3)
                d /= np.linalg.norm(d)
                r = np.linalg.norm(x[v - 1])
                if r > R:                    # pulled back into the globule
                    d = d - 1.5 * x[v - 1] / r
                    d /= np.linalg.norm(d)
                cand = x[v - 1] + step * d
                dmin = np.min(np.linalg.norm(x[:v] - cand, axis=1))
                if dmin > best_d:
                    best, best_d = cand, dmin
                if dmin > 1.0:
                    break
            x[v] = best
        g = nx.Graph()
        for v in range(n):
            g.add_node(v, element=int(Z[v]))
        D = np.linalg.norm(x[:, None, :] - x[None, :, :], axis=2)
        ii, jj = np.nonzero(np.triu(D < cutoff, k=1))
        for i, j in zip(ii.tolist(), jj.tolist()):
            d = float(np.float32(D[i, j]))
            g.add_edge(i, j, w=float(np.float32(1.0 - d / cutoff)), length=d)
        out.append(Graph.from_networkx(g, weight='w'))
    return Graph.unify_datatype(out)
