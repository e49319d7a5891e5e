"""Python face of the CPU oracle -- TEST INFRASTRUCTURE ONLY (see
mgk_oracle.c).  Nothing under graphdot_amd/ imports this module.

Two independent restatements of the reference's math are offered:

* ``mode='dense'``: fp64 dense assembly of the product-graph system and a
  direct ``numpy.linalg.solve`` -- the construction of the reference's own
  test oracle ``MLGK`` (/root/reference/test/kernel/marginalized/test_kernel.py:20-68)
  generalised to cross pairs the way ``M3._mlgk`` does
  (/root/reference/graphdot/experimental/metric/m3.py:52-106).
* ``mode='pcg32'`` / ``'pcg64'``: the C restatement of the device algorithm
  (Jacobi-PCG with the reference's stopping rules) in mgk_oracle.c.

Microkernels are evaluated through their *Python* ``__call__`` here, so the
oracle is also independent of the C++ expression generator.

``gram`` / ``diag`` assemble outputs exactly as
/root/reference/graphdot/kernel/marginalized/template.cu:123-205,226-469 writes
them (symmetric mirroring, nodal blocks, lmin, Jacobian column order).
"""
import ctypes
import os
import subprocess
import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build(quiet=True):
    """Compile the C oracle (gcc) if it is not there yet."""
    out = os.path.join(_here, '_build', 'libmgk_oracle.so')
    src = os.path.join(_here, 'mgk_oracle.c')
    if (not os.path.exists(out)
            or os.path.getmtime(out) < os.path.getmtime(src)):
        subprocess.run(['make', '-C', _here, '-s'], check=True,
                       stdout=subprocess.DEVNULL if quiet else None)
    return out


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the
    container's CPU quota (the GPU box shows 256 CPUs and grants 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def lib(omp=False):
    key = 'omp' if omp else 'seq'
    if key not in _libs:
        build()
        if omp:                        # (libgomp reads it when it is loaded)
            os.environ.setdefault('OMP_NUM_THREADS', str(usable_cpus()))
        name = 'libmgk_oracle_omp.so' if omp else 'libmgk_oracle.so'
        _libs[key] = ctypes.CDLL(os.path.join(_here, '_build', name))
    return _libs[key]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------
# graph -> arrays (reference: _octilegraph.py:107-157)
# --------------------------------------------------------------------------
def _row64(row):
    """The row with its numpy scalars as Python numbers: the microkernels are
    evaluated in float64 whatever type the table stores an attribute in.
    (The frames keep floats as float32, and numpy -- NEP 50 -- keeps float32
    arithmetic float32 when the other operand is a Python float: kernel values
    6e-8 off, which a double build of the solver is held to 1e-9 against;
    found by scripts/fuzz_parity.py.)"""
    def wide(v):
        if isinstance(v, np.floating):
            return float(v)
        if isinstance(v, np.integer):
            return int(v)
        if isinstance(v, np.ndarray) and v.dtype.kind == 'f':
            return v.astype(np.float64)
        return v
    return type(row)(*[wide(v) for v in tuple(row)])


class PairSide:
    """Directed-nonzero view of one graph: both orientations of every edge,
    a self loop once; degree = sum of incident weights (self loop once),
    isolated nodes get degree 1."""

    def __init__(self, g, wide=False):
        self.g = g
        self.n = n = len(g.nodes)
        order = np.argsort(np.asarray(g.nodes['!i']))
        widen = _row64 if wide else (lambda r: r)
        rows = [widen(r) for r in g.nodes.rows()]
        self.node_rows = [rows[k] for k in order]       # indexed by node id
        self.nodes_sorted = g.nodes[[c for c in g.nodes.columns]]
        ei = np.asarray(g.edges['!i']).astype(np.int64)
        ej = np.asarray(g.edges['!j']).astype(np.int64)
        self.weighted = '!w' in g.edges
        w = (np.asarray(g.edges['!w']).astype(np.float64) if self.weighted
             else np.ones(len(ei)))
        erows = [widen(r) for r in g.edges.rows()]
        di, dj, dw, dr = [], [], [], []
        deg = np.zeros(n)
        for k in range(len(ei)):
            i, j = int(ei[k]), int(ej[k])
            di.append(i); dj.append(j); dw.append(w[k]); dr.append(erows[k])
            deg[i] += w[k]
            if i != j:
                di.append(j); dj.append(i); dw.append(w[k])
                dr.append(erows[k])
                deg[j] += w[k]
        deg[deg == 0] = 1.0
        self.ei = np.array(di, dtype=np.int32)
        self.ej = np.array(dj, dtype=np.int32)
        self.ew = np.array(dw, dtype=np.float64)
        self.edge_rows = dr
        self.deg = deg
        self.nnz = len(di)


#: evaluate the microkernels on float64 copies of the attributes (`_row64`)?
#: Off by default: the golden vectors of tests/golden/ were recorded from the
#: reference's Python solver, which evaluates them in the attributes' storage
#: type, and the oracle is pinned to them at 1e-11.  On (`with wide_rows():`)
#: for holding a DOUBLE build of the solver to 1e-9 on attributes that
#: float32 arithmetic does not evaluate exactly.
WIDE_ROWS = False


class wide_rows:
    """Context: microkernels evaluated in float64 (see WIDE_ROWS)."""

    def __enter__(self):
        global WIDE_ROWS
        self.was, WIDE_ROWS = WIDE_ROWS, True

    def __exit__(self, *exc):
        global WIDE_ROWS
        WIDE_ROWS = self.was


def _side(g):
    key = 'oracle_side64' if WIDE_ROWS else 'oracle_side'
    try:
        return g.cookie[key]
    except KeyError:
        s = g.cookie[key] = PairSide(g, wide=WIDE_ROWS)
        return s


def node_table(knode, s1, s2, jac=False):
    V = np.empty((s1.n, s2.n))
    dV = None
    for a, r1 in enumerate(s1.node_rows):
        for b, r2 in enumerate(s2.node_rows):
            if jac:
                f, j = knode(r1, r2, jac=True)
                j = np.atleast_1d(np.asarray(j, dtype=float)).ravel()
                if dV is None:
                    dV = np.zeros((len(j), s1.n, s2.n))
                V[a, b] = f
                dV[:, a, b] = j
            else:
                V[a, b] = knode(r1, r2)
    return (V, dV) if jac else V


def edge_table(kedge, s1, s2, jac=False):
    """kappa_e(e1, e2) * w1 * w2 on all pairs of directed nonzeros (the
    reference wraps the edge kernel as TensorProduct(weight=Product(),
    label=kedge) for weighted graphs: _backend_cuda.py:274-276)."""
    E = np.empty((s1.nnz, s2.nnz))
    dE = None
    cache = {}
    for a, r1 in enumerate(s1.edge_rows):
        for b, r2 in enumerate(s2.edge_rows):
            key = (id(r1), id(r2))
            if key not in cache:
                if jac:
                    f, j = kedge(r1, r2, jac=True)
                    j = np.atleast_1d(np.asarray(j, dtype=float)).ravel()
                    cache[key] = (f, j)
                else:
                    cache[key] = (kedge(r1, r2), None)
            f, j = cache[key]
            w = s1.ew[a] * s2.ew[b]
            E[a, b] = f * w
            if jac:
                if dE is None:
                    dE = np.zeros((len(j), s1.nnz, s2.nnz))
                dE[:, a, b] = j * w
    if jac and dE is None:
        dE = np.zeros((0, s1.nnz, s2.nnz))
    return (E, dE) if jac else E


# --------------------------------------------------------------------------
# single-pair solvers
# --------------------------------------------------------------------------
def assemble(s1, s2, V, E, q):
    """Dense A = diag(Dx/Vx) - W and Dx."""
    n1, n2 = s1.n, s2.n
    N = n1 * n2
    Dx = np.kron(s1.deg, s2.deg) / (1 - q)**2
    A = np.diag(Dx / V.ravel())
    rows = (s1.ei[:, None] * n2 + s2.ei[None, :]).ravel()
    cols = (s1.ej[:, None] * n2 + s2.ej[None, :]).ravel()
    np.subtract.at(A, (rows, cols), E.ravel())
    assert A.shape == (N, N)
    return A, Dx


def solve_pair(s1, s2, V, E, q, mode='dense', tol=1e-8, rhs_extra=None,
               warm=None, q0=None):
    """Solution(s) of the pair system as (n1, n2) arrays.

    Returns (x, y, iters): x solves A x = Dx q^2/q0^2; y solves A y = rhs_extra
    (if given, else None)."""
    q0 = q if q0 is None else q0
    n1, n2 = s1.n, s2.n
    N = n1 * n2
    if mode == 'dense':
        A, Dx = assemble(s1, s2, V, E, q)
        x = np.linalg.solve(A, Dx * q * q / (q0 * q0))
        y = np.linalg.solve(A, rhs_extra.ravel()) if rhs_extra is not None \
            else None
        return (x.reshape(n1, n2),
                None if y is None else y.reshape(n1, n2), 0)
    real, sfx = {'pcg32': (np.float32, 'f32'),
                 'pcg64': (np.float64, 'f64')}[mode]
    creal = ctypes.c_float if real is np.float32 else ctypes.c_double
    L = lib()
    d1 = s1.deg.astype(real); d2 = s2.deg.astype(real)
    Vr = np.ascontiguousarray(V, dtype=real)
    Er = np.ascontiguousarray(E, dtype=real)
    common = (n1, n2, _ptr(d1), _ptr(d2), _ptr(Vr), s1.nnz, _ptr(s1.ei),
              _ptr(s1.ej), s2.nnz, _ptr(s2.ei), _ptr(s2.ej), _ptr(Er))
    if rhs_extra is None:
        x = np.zeros(N, dtype=real)
        if warm is not None:
            x[:] = np.asarray(warm, dtype=real).ravel()
        work = np.empty(4 * N, dtype=real)
        f = getattr(L, f'mgk_pcg_{sfx}')
        f.restype = ctypes.c_int
        k = f(*common, creal(q), creal(q0), creal(tol),
              int(warm is not None), _ptr(x), _ptr(work))
        return x.reshape(n1, n2).astype(np.float64), None, k
    x = np.zeros(2 * N, dtype=real)
    work = np.empty(8 * N, dtype=real)
    px = np.ascontiguousarray(rhs_extra, dtype=real).ravel()
    f = getattr(L, f'mgk_pcg_duo_{sfx}')
    f.restype = ctypes.c_int
    k = f(*common, _ptr(px), creal(q), creal(q0), _ptr(x), _ptr(work))
    x = x.astype(np.float64)
    return x[:N].reshape(n1, n2), x[N:].reshape(n1, n2), k


def derivative(s1, s2, V, dV, dE, p1, p2, dp1, dp2, q, x, y):
    """Analytic Jacobian [p..., q, node..., edge...] through the C
    restatement of marginalized_kernel.h:806-997 (fp64)."""
    L = lib()
    n1, n2 = s1.n, s2.n
    N = n1 * n2
    c = np.ascontiguousarray
    d1 = c(s1.deg, dtype=np.float64); d2 = c(s2.deg, dtype=np.float64)
    Vr = c(V, dtype=np.float64)
    xs = np.concatenate((x.ravel(), y.ravel())).astype(np.float64)
    np_, nv, ne = dp1.shape[0], dV.shape[0], dE.shape[0]
    jac = np.zeros(np_ + 1 + nv + ne)
    p1 = c(p1, dtype=np.float64); p2 = c(p2, dtype=np.float64)
    dp1 = c(dp1, dtype=np.float64); dp2 = c(dp2, dtype=np.float64)
    dVr = c(dV, dtype=np.float64); dEr = c(dE, dtype=np.float64)
    L.mgk_derivative_f64(
        n1, n2, _ptr(d1), _ptr(d2), _ptr(Vr), s1.nnz, _ptr(s1.ei),
        _ptr(s1.ej), s2.nnz, _ptr(s2.ei), _ptr(s2.ej), _ptr(p1), _ptr(p2),
        np_, _ptr(dp1), _ptr(dp2), nv, _ptr(dVr), ne, _ptr(dEr),
        ctypes.c_double(q), _ptr(xs), _ptr(jac))
    assert N == x.size
    return jac


# --------------------------------------------------------------------------
# API-level oracle: same outputs as MarginalizedGraphKernel.__call__ / diag
# --------------------------------------------------------------------------
def _start_prob(p, s):
    """p: number or StartingProbability-like (callable on the node frame)."""
    if np.isscalar(p):
        return np.full(s.n, float(p)), np.ones((1, s.n))
    pv, dp = p(s.g.nodes)
    order = np.argsort(np.asarray(s.g.nodes['!i']))
    pv = np.asarray(pv, dtype=float)[order]
    dp = np.asarray(dp, dtype=float)
    dp = dp[:, order] if dp.size else np.zeros((0, s.n))
    return pv, dp


def _flat_theta(kernel):
    from graphdot_amd.util.iterable import flatten
    return np.array(list(flatten(kernel.theta)), dtype=float)


def _with_theta(kernel, flat):
    import copy
    from graphdot_amd.util.iterable import fold_like
    k = copy.deepcopy(kernel)
    k.theta = fold_like(list(flat), k.theta)
    return k


def pair_value(g1, g2, knode, kedge, p=1.0, q=0.01, lmin=0, mode='dense',
               tol=1e-8, eval_gradient=False, nodal=False, eps=1e-2):
    """One pair: returns (R, dR) where R is the (n1, n2) nodal matrix
    x * p1 * p2 (after the lmin correction) when `nodal`, else its sum.
    dR follows template.cu: analytic for graph-level (:422-469), central
    finite differences in log-theta for nodal outputs (:226-418)."""
    s1, s2 = _side(g1), _side(g2)
    p1, dp1 = _start_prob(p, s1)
    p2, dp2 = _start_prob(p, s2)
    V, dV = node_table(knode, s1, s2, jac=True) if eval_gradient else \
        (node_table(knode, s1, s2), None)
    if eval_gradient:
        E, dE = edge_table(kedge, s1, s2, jac=True)
    else:
        E, dE = edge_table(kedge, s1, s2), None
    px = np.outer(p1, p2)

    def post(x, Vtab):
        return x - Vtab if lmin == 1 else x

    if not eval_gradient:
        x, _, _ = solve_pair(s1, s2, V, E, q, mode, tol)
        R = post(x, V) * px
        return (R if nodal else R.sum()), None

    if nodal is False:
        x, y, _ = solve_pair(s1, s2, V, E, q, mode, tol, rhs_extra=px)
        # (explicit widths: a microkernel without hyperparameters has zero
        # planes, and numpy cannot infer -1 for an empty array)
        jac = derivative(s1, s2, V, dV.reshape(dV.shape[0], V.size),
                         dE.reshape(dE.shape[0], E.size), p1, p2, dp1, dp2, q,
                         x, y)
        return (post(x, V) * px).sum(), jac

    # nodal gradient: finite differences with the *uncorrected* solution
    # (template.cu: lmin is applied to x only, not to the FD re-solves)
    x, _, _ = solve_pair(s1, s2, V, E, q, mode, tol)
    R = post(x, V) * px
    cols = []
    for j in range(dp1.shape[0]):
        cols.append(x * (p1[:, None] * dp2[j][None, :]
                         + p2[None, :] * dp1[j][:, None]))
    gt = tol if mode == 'dense' else 1e-6

    def fd(Vp, Ep, qp, Vm, Em, qm, denom):
        xp, _, _ = solve_pair(s1, s2, Vp, Ep, qp, mode, gt)
        xm, _, _ = solve_pair(s1, s2, Vm, Em, qm, mode, gt)
        return (xp - xm) / denom * px

    cols.append(fd(V, E, np.exp(np.log(q) + eps), V, E,
                   np.exp(np.log(q) - eps), 2 * eps * q))
    tv = _flat_theta(knode)
    for j in range(len(tv)):
        tp, tm = tv.copy(), tv.copy()
        tp[j] = np.exp(np.log(tv[j]) + eps)
        tm[j] = np.exp(np.log(tv[j]) - eps)
        cols.append(fd(node_table(_with_theta(knode, tp), s1, s2), E, q,
                       node_table(_with_theta(knode, tm), s1, s2), E, q,
                       2 * eps * tv[j]))
    te = _flat_theta(kedge)
    for j in range(len(te)):
        tp, tm = te.copy(), te.copy()
        tp[j] = np.exp(np.log(te[j]) + eps)
        tm[j] = np.exp(np.log(te[j]) - eps)
        cols.append(fd(V, edge_table(_with_theta(kedge, tp), s1, s2), q,
                       V, edge_table(_with_theta(kedge, tm), s1, s2), q,
                       2 * eps * te[j]))
    return R, np.stack(cols, axis=-1)


def gram(X, knode, kedge, Y=None, p=1.0, q=0.01, lmin=0, nodal=False,
         eval_gradient=False, mode='dense', tol=1e-8, eps=1e-2):
    """Oracle for MarginalizedGraphKernel.__call__ (all hyperparameters
    active; the caller masks columns)."""
    sym = Y is None
    Yl = X if sym else Y
    nx, ny = len(X), len(Yl)
    if nodal:
        sx = np.concatenate(([0], np.cumsum([len(g.nodes) for g in X])))
        sy = np.concatenate(([0], np.cumsum([len(g.nodes) for g in Yl])))
    else:
        sx, sy = np.arange(nx + 1), np.arange(ny + 1)
    K = np.zeros((sx[-1], sy[-1]))
    dK = None
    for a in range(nx):
        for b in range(a if sym else 0, ny):
            R, J = pair_value(X[a], Yl[b], knode, kedge, p, q, lmin, mode,
                              tol, eval_gradient, nodal, eps)
            K[sx[a]:sx[a + 1], sy[b]:sy[b + 1]] = R
            if sym and a != b:
                K[sy[b]:sy[b + 1], sx[a]:sx[a + 1]] = np.transpose(R)
            if eval_gradient:
                J = np.asarray(J)
                if dK is None:
                    dK = np.zeros(K.shape + (J.shape[-1],))
                dK[sx[a]:sx[a + 1], sy[b]:sy[b + 1], :] = J
                if sym and a != b:
                    dK[sy[b]:sy[b + 1], sx[a]:sx[a + 1], :] = (
                        np.swapaxes(J, 0, 1) if nodal else J)
    return (K, dK) if eval_gradient else K


def diag(X, knode, kedge, p=1.0, q=0.01, lmin=0, nodal=False,
         eval_gradient=False, mode='dense', tol=1e-8, eps=1e-2):
    """Oracle for MarginalizedGraphKernel.diag."""
    vals, grads = [], []
    for g in X:
        R, J = pair_value(g, g, knode, kedge, p, q, lmin, mode, tol,
                          eval_gradient, nodal is not False, eps)
        if nodal is True:
            vals.append(np.diag(R))
            if eval_gradient:
                grads.append(np.stack([np.diag(J[..., k])
                                       for k in range(J.shape[-1])], -1))
        elif nodal == 'block':
            vals.append(R)
        else:
            vals.append(np.atleast_1d(R))
            if eval_gradient:
                grads.append(np.asarray(J)[None, :])
    if nodal == 'block':
        return vals
    out = np.concatenate(vals)
    return (out, np.concatenate(grads, axis=0)) if eval_gradient else out


# --------------------------------------------------------------------------
# Batched tensor-product path in C (CPU baseline / full-size checker)
# --------------------------------------------------------------------------
class _CGraph(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int), ('nnz', ctypes.c_int),
                ('nodef', ctypes.c_void_p), ('ei', ctypes.c_void_p),
                ('ej', ctypes.c_void_p), ('edgef', ctypes.c_void_p),
                ('ew', ctypes.c_void_p), ('deg', ctypes.c_void_p)]


_TP_TYPES = {'Constant': 0, 'KroneckerDelta': 1, 'SquareExponential': 2}


def _tp_spec(kernel):
    """(feature names, type codes, params) of a TensorProduct of elementary
    microkernels, or of a bare Constant (no features)."""
    if kernel.name == 'Constant':
        return [None], np.array([0], np.int32), np.array([kernel.c], float)
    if kernel.name != 'Composite' or kernel.opstr != '*':
        raise ValueError('batched oracle handles TensorProduct/Constant only')
    names, types, params = [], [], []
    for key, k in kernel.kw_kernels.items():
        if k.name not in _TP_TYPES:
            raise ValueError(f'unsupported elementary kernel {k.name}')
        names.append(key)
        types.append(_TP_TYPES[k.name])
        params.append(float(list(k.theta)[0]))
    return names, np.array(types, np.int32), np.array(params, float)


class TensorProductBatch:
    """Packs graphs once for ``mgk_gram_tp_*`` (mgk_oracle.c)."""

    def __init__(self, graphs, knode, kedge):
        self.vnames, self.vtype, self.vparam = _tp_spec(knode)
        self.enames, self.etype, self.eparam = _tp_spec(kedge)
        self._keep = []
        arr = (_CGraph * len(graphs))()
        for k, g in enumerate(graphs):
            s = _side(g)
            order = np.argsort(np.asarray(g.nodes['!i']))
            nodef = np.ascontiguousarray(np.column_stack([
                np.zeros(s.n) if c is None
                else np.asarray(g.nodes[c], dtype=float)[order]
                for c in self.vnames]))
            # per directed nonzero edge features
            eidx = []
            for e, (i, j) in enumerate(zip(g.edges['!i'], g.edges['!j'])):
                eidx.append(e)
                if i != j:
                    eidx.append(e)
            eidx = np.array(eidx, dtype=int)
            edgef = np.ascontiguousarray(np.column_stack([
                np.zeros(s.nnz) if c is None
                else np.asarray(g.edges[c], dtype=float)[eidx]
                for c in self.enames]))
            keep = (nodef, edgef, s.ei, s.ej, s.ew, s.deg)
            self._keep.append(keep)
            arr[k] = _CGraph(s.n, s.nnz, nodef.ctypes.data, s.ei.ctypes.data,
                             s.ej.ctypes.data, edgef.ctypes.data,
                             s.ew.ctypes.data, s.deg.ctypes.data)
        self.arr = arr

    def run(self, ji, jj, p=1.0, q=0.01, tol=1e-8, lmin=0, real='f32',
            omp=False):
        ji = np.ascontiguousarray(ji, dtype=np.int32)
        jj = np.ascontiguousarray(jj, dtype=np.int32)
        out = np.zeros(len(ji))
        iters = np.zeros(len(ji), dtype=np.int32)
        f = getattr(lib(omp), f'mgk_gram_tp_{real}')
        f.restype = ctypes.c_int
        rc = f(len(ji), _ptr(ji), _ptr(jj), self.arr, len(self.vtype),
               _ptr(self.vtype), _ptr(self.vparam), len(self.etype),
               _ptr(self.etype), _ptr(self.eparam), ctypes.c_double(p),
               ctypes.c_double(q), ctypes.c_double(tol), int(lmin),
               _ptr(out), _ptr(iters))
        if rc:
            raise MemoryError('oracle allocation failed')
        return out, iters

    def run_gradient(self, ji, jj, p=1.0, q=0.01, lmin=0, real='f64',
                     omp=False):
        """Value and analytic Jacobian per job (``mgk_gram_tp_grad_*``:
        compute_duo + derivative).  Columns: [p, q, node hyperparameters...,
        edge hyperparameters...] -- all of them, the caller masks."""
        ji = np.ascontiguousarray(ji, dtype=np.int32)
        jj = np.ascontiguousarray(jj, dtype=np.int32)
        nJ = 2 + len(self.vtype) + len(self.etype)
        out = np.zeros(len(ji))
        grad = np.zeros((len(ji), nJ))
        iters = np.zeros(len(ji), dtype=np.int32)
        f = getattr(lib(omp), f'mgk_gram_tp_grad_{real}')
        f.restype = ctypes.c_int
        rc = f(len(ji), _ptr(ji), _ptr(jj), self.arr, len(self.vtype),
               _ptr(self.vtype), _ptr(self.vparam), len(self.etype),
               _ptr(self.etype), _ptr(self.eparam), ctypes.c_double(p),
               ctypes.c_double(q), int(lmin), _ptr(out), _ptr(grad),
               _ptr(iters))
        if rc:
            raise MemoryError('oracle allocation failed')
        return out, grad, iters
