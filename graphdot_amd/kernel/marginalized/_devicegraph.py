"""Host-side packer of the device graph format (``csrc/device/graph.h``).

Takes the role of the reference's ``OctileGraph``
(``graphdot/kernel/marginalized/_octilegraph.py:37-177``) and keeps its
*semantics* -- AoS node labels indexed by node id, fp32 degrees as sums of
incident weights with a self loop counted once and ``0 -> 1``, both
orientations of every edge, ``{weight, label}`` edge structs for weighted
graphs, phantom ``labeled`` fields for unlabeled graphs, variable-length
attributes as ``frozen_array`` views -- but not its 8x8 octile tiling: the
MI355X solver consumes a flat list of directed nonzeros (see ``graph.h`` for
why and for the ordering rule).

Each graph packs into one relocatable byte blob::

    [degree f32[n]] [nodes node_t[n]] [rowptr u16[n+1]] [nz u16x2[nnz]]
    [edges edge_t[nnz]] [perm u16[n]] [variable-length attribute payloads ...]

Nodes are renumbered by descending adjacency count (``perm[new] = original
id``); the directed nonzeros are stored in CSR order of the new numbering.
The solver's degree-batched mat-vec relies on that order (mgk_solver.h).

with every section 16-byte aligned; pointers inside the blob (frozen_array
views) are stored as blob-relative offsets plus a relocation list, so blobs of
many graphs are concatenated into one arena and uploaded with a single copy.
"""
import itertools as it
import numpy as np
from operator import itemgetter
from numpy.lib.recfunctions import repack_fields
from ...codegen.cpptool import cpptype
from ...codegen.typetool import common_min_type, is_scalar_type

_ALIGN = 16

#: device-side header, mirrors graphdot::graph_header_t (64 bytes); the
#: section fields are byte offsets from the arena base, `hist[d]` counts the
#: nodes with d adjacency nonzeros (hist[15]: 15 or more)
HEADER_DTYPE = np.dtype([
    ('n_node', np.int32), ('n_nz', np.int32), ('degree', np.uint32),
    ('node', np.uint32), ('rowptr', np.uint32), ('nz', np.uint32),
    ('edge', np.uint32), ('perm', np.uint32), ('hist', np.uint16, (16,))],
    align=True)
assert HEADER_DTYPE.itemsize == 64
HIST_BINS = 16


def degree_histogram(adjacency_count):
    """hist field of the header: nodes per adjacency count, the last bin
    open-ended."""
    c = np.minimum(np.asarray(adjacency_count, dtype=np.int64), HIST_BINS - 1)
    return np.bincount(c, minlength=HIST_BINS).astype(np.uint16)
SECTIONS = ('degree', 'node', 'rowptr', 'nz', 'edge', 'perm')

NZ_DTYPE = np.dtype([('i', np.uint16), ('j', np.uint16)])


def degree_histograms(dgraphs):
    """(n_graphs, 16) degree histograms of packed graphs in one pass."""
    n = len(dgraphs)
    hist = np.zeros((n, HIST_BINS), dtype=np.int64)
    if n:
        n_node = np.array([g.n_node for g in dgraphs], dtype=np.int64)
        gid = np.repeat(np.arange(n), n_node)
        cnt = np.minimum(np.concatenate(
            [g.adjacency_count for g in dgraphs]), HIST_BINS - 1)
        np.add.at(hist, (gid, cnt), 1)
    return hist


def whole_batch(dgraphs):
    """The packed batch (`pack_many`'s result dict) that `dgraphs` is, graph
    for graph and in order -- or None."""
    hit = getattr(dgraphs, 'batch', False)
    if hit is not False:
        return hit
    n = len(dgraphs)
    b0 = getattr(dgraphs[0], '_b', None) if n else None
    if b0 is not None and not (
            len(b0['sec_off']) == n and all(
                getattr(g, '_b', None) is b0 and g._k == k
                for k, g in enumerate(dgraphs))):
        b0 = None
    try:
        dgraphs.batch = b0
    except AttributeError:                 # a plain list
        pass
    return b0


def graph_features(dgraphs):
    """Per-graph quantities of a list of packed graphs as int64 arrays:
    n_node, n_nz, image_bytes, max_degree and the 16-bin degree histogram
    hist (n, 16).  A whole natively packed batch in its order hands out the
    arrays made at pack time; any other list is walked graph by graph.  Kept
    on the list when it can hold attributes (the backend's list of a call)."""
    hit = getattr(dgraphs, 'features', None)
    if hit is not None:
        return hit
    b0 = whole_batch(dgraphs)
    if b0 is not None:
        out = b0['features']
    else:
        out = dict(
            n_node=np.array([g.n_node for g in dgraphs], dtype=np.int64),
            n_nz=np.array([g.n_nz for g in dgraphs], dtype=np.int64),
            image_bytes=np.array([g.image_bytes for g in dgraphs],
                                 dtype=np.int64),
            max_degree=np.array([g.max_degree for g in dgraphs],
                                dtype=np.int64),
            hist=degree_histograms(dgraphs))
    try:
        dgraphs.features = out
    except AttributeError:                 # a plain list
        pass
    return out


@cpptype(ptr=np.intp, size=np.int32)
class FrozenArray(np.ndarray):
    """An ndarray slice that packs as {pointer, length}; the pointer is a
    blob-relative offset until the arena is relocated."""
    _offset = 0

    @property
    def ptr(self):
        return self._offset

    @property
    def size(self):
        return len(self)


def _pad(n, a=_ALIGN):
    return (n + a - 1) // a * a


def _widen(dtype, real):
    """Map float32 attribute fields to `real` (fp64 builds compute on the
    attributes in double precision)."""
    if real == np.float32:
        return dtype
    dtype = np.dtype(dtype)
    if dtype.names is not None:
        return np.dtype([(k, _widen(dtype.fields[k][0], real))
                         for k in dtype.names], align=True)
    return np.dtype(real) if dtype == np.float32 else dtype


class DeviceGraph:
    """Packed image of one :py:class:`graphdot_amd.graph.Graph`.

    Attributes
    ----------
    node_t, edge_t: numpy aligned struct dtypes (the C++ node/edge types)
    weighted: bool
    n_node, n_nz: int
    perm: uint16[n_node], perm[new id] = original node id
    degree: float32[n_node] (new numbering)
    rowptr, nz: CSR of the directed nonzeros (new numbering)
    blob: uint8 array, relocatable image
    offsets: dict section -> byte offset in blob
    relocs: byte offsets (in blob) of uint64 words that hold blob-relative
        pointers
    """

    def __init__(self, graph, real=np.float32):
        real = np.dtype(real).type
        nodes = graph.nodes.copy(deep=False)
        edges = graph.edges.copy(deep=False)
        self.n_node = n = len(nodes)
        if n > 0xFFFF or 2 * len(edges) > 0xFFFF:
            raise ValueError(
                f'graph with {n} nodes and {2 * len(edges)} directed '
                'adjacency nonzeros: the device format indexes both with 16 '
                'bits (at most 65535 each)')

        varlen = []     # (payload ndarray) in blob order
        for df in (nodes, edges):
            for key in list(df.columns):
                col = df[key]
                if is_scalar_type(col.dtype):
                    continue
                if col.concrete_type not in (list, tuple, np.ndarray) and \
                        not (isinstance(col.concrete_type, type) and
                             issubclass(col.concrete_type,
                                        (list, tuple, np.ndarray))):
                    raise TypeError(
                        f'Unsupported non-scalar attribute {key} of type '
                        f'{col.concrete_type}')
                inner = common_min_type.of_types(
                    [x.dtype if isinstance(x, np.ndarray)
                     else common_min_type.of_values(x) for x in col])
                if not is_scalar_type(inner):
                    raise TypeError(
                        'List-like graph attributes must have scalar '
                        f'elements. Attribute {key} is {inner}.')
                inner = np.dtype(_widen(np.dtype(inner), real))
                flat = np.fromiter(it.chain.from_iterable(col), dtype=inner)
                sizes = np.fromiter(map(len, col), dtype=np.int64,
                                    count=len(col))
                heads = np.cumsum(sizes) - sizes
                views = np.empty(len(col), dtype=object)
                for k, (h, s) in enumerate(zip(heads, sizes)):
                    v = flat[h:h + s].view(FrozenArray)
                    v._payload = len(varlen)
                    v._head = int(h) * inner.itemsize
                    views[k] = v
                varlen.append(flat)
                tag = f'${key}::frozen_array::{inner.str}'
                df[tag] = views
                df.drop([key], inplace=True)

        # phantom labels keep node_t / edge_t non-empty
        if len(nodes.columns) == 1:
            nodes['labeled'] = np.zeros(n, np.bool_)
        if len(edges.columns) == 2:     # only !i, !j: unweighted, unlabeled
            edges['labeled'] = np.zeros(len(edges), np.bool_)

        # ---- directed nonzeros, degrees --------------------------------------
        idx = np.asarray(nodes['!i']).astype(np.int64)
        nodes.drop(['!i'], inplace=True)
        ei = np.asarray(edges['!i']).astype(np.int64)
        ej = np.asarray(edges['!j']).astype(np.int64)
        m = len(ei)
        self.weighted = '!w' in edges
        w = (np.asarray(edges['!w']).astype(np.float32) if self.weighted
             else np.ones(m, np.float32))
        degree = np.zeros(n, np.float32)
        np.add.at(degree, ei, w)
        np.add.at(degree, ej, w)
        loops = ei == ej
        np.subtract.at(degree, ei[loops], w[loops])
        degree[degree == 0] = 1.0

        # both orientations; duplicates (self loops, repeated edges) collapse
        # onto their first occurrence like the reference's np.unique
        src = np.concatenate((ei, ej))
        dst = np.concatenate((ej, ei))
        eid = np.concatenate((np.arange(m), np.arange(m)))
        _, first = np.unique(src * n + dst, return_index=True)
        src, dst, eid = src[first], dst[first], eid[first]

        # renumber nodes by descending adjacency count (stable)
        count = np.bincount(src, minlength=n)
        perm = np.argsort(-count, kind='stable')        # new -> original
        rank = np.empty(n, dtype=np.int64)
        rank[perm] = np.arange(n)                       # original -> new
        self.perm = perm.astype(np.uint16)
        self.rank = rank
        self.degree = degree = degree[perm]
        self.adjacency_count = count[perm]
        src, dst = rank[src], rank[dst]
        order = np.lexsort((dst, src))                  # CSR order
        src, dst, eid = src[order], dst[order], eid[order]
        self.n_nz = nnz = len(src)
        self.nz = np.zeros(nnz, dtype=NZ_DTYPE)
        self.nz['i'], self.nz['j'] = src, dst
        self.rowptr = np.concatenate(
            ([0], np.cumsum(count[perm]))).astype(np.uint16)
        self.edge_index = eid

        # ---- nodes: AoS indexed by the *new* node id ---------------------------
        self.node_t = node_t = _widen(nodes.rowtype(), real)
        nodes_aos = np.zeros(n, dtype=node_t)
        node_fa = []          # (row, field, FrozenArray)
        self._fill(nodes_aos, rank[idx], nodes, node_t, node_fa)

        # ---- edges ---------------------------------------------------------------
        label_df = edges.drop(['!i', '!j', '!w'])
        label_t = _widen(label_df.rowtype(), real)
        if self.weighted:
            edge_t = np.dtype([('weight', real), ('label', label_t)],
                              align=True)
        else:
            edge_t = label_t
        self.edge_t = edge_t
        # cheap equality token for "same node / edge record types"
        self.signature = (self.weighted, str(self.node_t), str(edge_t))
        edges_aos = np.zeros(nnz, dtype=edge_t)
        edge_fa = []
        target = edges_aos['label'] if self.weighted and label_t.itemsize \
            else edges_aos
        if self.weighted:
            edges_aos['weight'] = w[eid]
        if label_t.itemsize:
            self._fill(target, np.arange(nnz), label_df, label_t, edge_fa,
                       take=eid,
                       base=edge_t.fields['label'][1] if self.weighted else 0)
            if self.weighted:
                edges_aos['label'] = target

        # ---- blob ---------------------------------------------------------------
        sections, cursor = {}, 0
        for name, arr in (('degree', degree), ('node', nodes_aos),
                          ('rowptr', self.rowptr), ('nz', self.nz),
                          ('edge', edges_aos), ('perm', self.perm)):
            sections[name] = (cursor, arr)
            cursor += _pad(arr.nbytes)
        payload_off = []
        for arr in varlen:
            payload_off.append(cursor)
            sections[f'payload{len(payload_off) - 1}'] = (cursor, arr)
            cursor += _pad(arr.nbytes)
        blob = np.zeros(max(cursor, _ALIGN), dtype=np.uint8)
        for off, arr in sections.values():
            blob[off:off + arr.nbytes] = arr.view(np.uint8).ravel() \
                if arr.nbytes else []
        self.offsets = {k: v[0] for k, v in sections.items()}
        #: bytes of the contiguous image [degree .. perm] that the solver
        #: stages in LDS (payloads of variable-length attributes excluded)
        self.image_bytes = _pad(self.offsets['perm'] + 2 * n)

        relocs = []
        for base_name, base_t, fa_list in (('node', node_t, node_fa),
                                           ('edge', edge_t, edge_fa)):
            for row, field_off, v in fa_list:
                where = (self.offsets[base_name] + row * base_t.itemsize
                         + field_off)
                word = blob[where:where + 8].view(np.uint64)
                word[0] = payload_off[v._payload] + v._head
                relocs.append(where)
        self.relocs = np.array(relocs, dtype=np.int64)
        self.blob = blob

    @staticmethod
    def _field_offset(dtype, name):
        return dtype.fields[name][1]

    def _fill(self, aos, index, frame, dtype, fa_out, take=None, base=0):
        """Scatter the columns of `frame` into struct array `aos` at rows
        `index`; frozen_array cells get {offset placeholder, size} and are
        recorded in `fa_out` as (row, byte offset of the pointer, view)."""
        for name in dtype.names:
            col = frame[name]
            if take is not None:
                col = np.asarray(col)[take] if is_scalar_type(col.dtype) \
                    else [col[k] for k in take]
            ft, foff = dtype.fields[name][0], dtype.fields[name][1]
            if name.startswith('$'):
                for row, v in zip(index, col):
                    aos[name]['size'][row] = len(v)
                    fa_out.append((int(row), base + foff
                                   + ft.fields['ptr'][1], v))
            else:
                aos[name][index] = np.asarray(col).astype(ft)

    # -- reference-compatible read-only views ---------------------------------
    @property
    def max_degree(self):
        """Largest adjacency count of a node (cached)."""
        try:
            return self._max_degree
        except AttributeError:
            self._max_degree = int(self.adjacency_count.max()) \
                if self.n_node else 0
            return self._max_degree

    @property
    def degree_hist(self):
        """hist field of the graph header (cached)."""
        try:
            return self._degree_hist
        except AttributeError:
            self._degree_hist = degree_histogram(self.adjacency_count)
            return self._degree_hist

    @property
    def state(self):
        raise AttributeError(
            'DeviceGraph has no absolute-address state; headers are produced '
            'by GraphArena once the blob has a device address.')


def _scalar_frame(df):
    """(column name, dtype) of every column if all of them are scalar."""
    out = []
    for key, col in df._data.items():
        kind = col.dtype.kind
        if kind not in 'biuf':          # is_scalar_type, inlined
            return None
        out.append((key, col.dtype.str))
    return tuple(out)


def pack_many(graphs, real=np.float32, native=True):
    """Pack a whole list of graphs in one pass: the native packer
    (`gdh_pack_graphs`, csrc/gdhost.cpp: degrees, directed nonzeros, degree
    renumbering, CSR and the blobs of all graphs in C++; numpy only
    concatenates the input tables and builds the AoS records), or with
    `native=False` the vectorised numpy restatement below -- the
    specification the native results are held to, byte for byte
    (tests/test_host_model.py).  The batched counterpart of the reference's
    per-graph ``OctileGraph.__init__``
    (``graphdot/kernel/marginalized/_octilegraph.py:37-177``).  Graphs with
    variable-length attributes, or whose tables differ in columns / types from
    the first graph's, are packed one by one.  Returns the list of DeviceGraph
    objects; their blobs are views into one shared buffer."""
    if not native:
        return pack_many_numpy(graphs, real)
    from ...hip import hostlib
    real = np.dtype(real).type
    graphs = list(graphs)
    if not graphs:
        return []
    sig_n, sig_e = _scalar_frame(graphs[0].nodes), _scalar_frame(graphs[0].edges)
    # same tables as the first graph's: by the row types the kernel's type
    # check left in the cookies (Graph.has_unified_types) when they are there
    # -- equal row types are equal column names and types -- else column by
    # column
    rt0 = graphs[0].cookie.get('rowtypes') if hasattr(
        graphs[0].cookie, 'get') else None

    # The tables of all graphs as flat columns in one native pass over the
    # objects (hostlib.collect_columns, csrc/gdcollect.cpp: it also checks
    # that every graph has the first one's columns and element types); any
    # irregularity -- or no extension module -- takes the Python path below.
    fast = None
    if sig_n is not None and sig_e is not None:
        f0n, f0e = graphs[0].nodes._data, graphs[0].edges._data
        cn = hostlib.collect_columns(graphs, 'nodes', list(f0n),
                                     {k: v.dtype for k, v in f0n.items()})
        ce = cn and hostlib.collect_columns(
            graphs, 'edges', list(f0e), {k: v.dtype for k, v in f0e.items()})
        if cn and ce and int(cn[1].max(initial=0)) <= 0xFFFF \
                and 2 * int(ce[1].max(initial=0)) <= 0xFFFF:
            fast = (cn, ce)

    def same(g):
        rt = g.cookie.get('rowtypes') if rt0 is not None else None
        if rt is not None:
            return (rt[0] is rt0[0] or rt[0] == rt0[0]) and \
                (rt[1] is rt0[1] or rt[1] == rt0[1])
        return _scalar_frame(g.nodes) == sig_n and \
            _scalar_frame(g.edges) == sig_e
    batch = list(range(len(graphs))) if fast else \
        [k for k, g in enumerate(graphs)
         if sig_n is not None and sig_e is not None and same(g)
         and len(g.nodes._data['!i']) <= 0xFFFF
         and 2 * len(g.edges._data['!i']) <= 0xFFFF]
    out = [None] * len(graphs)
    for k in sorted(set(range(len(graphs))) - set(batch)):
        out[k] = DeviceGraph(graphs[k], real=real)
    if not batch:
        return out
    gs = [graphs[k] for k in batch]
    G = len(gs)
    nframes, eframes = [g.nodes for g in gs[:1]], [g.edges for g in gs[:1]]

    def columns(frames):
        """{column: the arrays of all frames} with one C-level lookup per
        frame (the frames have the first one's columns: `same` above)"""
        keys = list(frames[0]._data)
        get = itemgetter(*keys)
        rows = [get(f._data) for f in frames]
        if len(keys) == 1:
            return {keys[0]: rows}
        return dict(zip(keys, zip(*rows)))

    if fast:
        (ncols, n), (ecols, m) = fast

        def cat(frames, key):
            return (ncols if frames is nframes else ecols)[key]
    else:
        nframes, eframes = [g.nodes for g in gs], [g.edges for g in gs]
        ncols, ecols = columns(nframes), columns(eframes)

        def cat(frames, key):
            return np.concatenate((ncols if frames is nframes else ecols)[key])

        n = np.fromiter(map(len, ncols['!i']), dtype=np.int64, count=G)
        m = np.fromiter(map(len, ecols['!i']), dtype=np.int64, count=G)
    node0 = np.concatenate(([0], np.cumsum(n)))
    edge0 = np.concatenate(([0], np.cumsum(m)))
    Nn, Ne = int(node0[-1]), int(edge0[-1])
    weighted = '!w' in eframes[0]
    w = cat(eframes, '!w').astype(np.float32) if weighted else None
    # record types from the first graph (phantom labels included)
    nodes0 = nframes[0].copy(deep=False)
    edges0 = eframes[0].copy(deep=False)
    if len(nodes0.columns) == 1:
        nodes0['labeled'] = np.zeros(len(nodes0), np.bool_)
    if len(edges0.columns) == 2:
        edges0['labeled'] = np.zeros(len(edges0), np.bool_)
    node_t = _widen(nodes0.drop(['!i']).rowtype(), real)
    label_t = _widen(edges0.drop(['!i', '!j', '!w']).rowtype(), real)
    edge_t = (np.dtype([('weight', real), ('label', label_t)], align=True)
              if weighted else label_t)
    # AoS records in input row order (the packer moves them into place)
    node_rec = np.zeros(Nn, dtype=node_t)
    for name in node_t.names:
        if name == 'labeled' and name not in nframes[0]:
            continue
        node_rec[name] = cat(nframes, name).astype(node_t.fields[name][0])
    label_rec = np.zeros(Ne, dtype=label_t)
    for name in label_t.names or ():
        if name == 'labeled' and name not in eframes[0]:
            continue
        label_rec[name] = cat(eframes, name).astype(label_t.fields[name][0])
    r = hostlib.pack_graphs(
        node0, edge0, cat(nframes, '!i'), cat(eframes, '!i'),
        cat(eframes, '!j'), w, node_rec, label_rec, edge_t.itemsize,
        edge_t.fields['label'][1] if weighted else 0,
        np.dtype(real).itemsize if weighted else 0)
    signature = (weighted, str(node_t), str(edge_t))
    r['nz'] = r['nz'].view(NZ_DTYPE)
    r['no_relocs'] = np.zeros(0, dtype=np.int64)
    r['blob_off'], r['nz_off'] = r['blob_off'].tolist(), r['nz_off'].tolist()
    r['node_off'] = node0.tolist()
    # per-graph quantities of the whole batch as arrays (graph_features)
    gid = np.repeat(np.arange(len(batch)), n)
    hist = np.zeros((len(batch), HIST_BINS), dtype=np.int64)
    np.add.at(hist, (gid, np.minimum(r['count'], HIST_BINS - 1)), 1)
    r['features'] = dict(
        n_node=np.asarray(n, dtype=np.int64),
        n_nz=np.asarray(r['nnz'], dtype=np.int64),
        image_bytes=_pad(r['sec_off'][:, 5] + 2 * np.asarray(n, np.int64)),
        max_degree=np.asarray(r['maxdeg'], dtype=np.int64), hist=hist)
    r['sec_off_list'] = offs = r['sec_off'].tolist()
    maxdeg = r['maxdeg'].tolist()
    nnz = r['nnz'].tolist()
    nl = n.tolist()
    # (a thousand small objects in a row: keep the cyclic collector from
    # walking the caller's heap in the middle of it -- nothing here can be
    # part of a cycle)
    from ...util import gcpause
    with gcpause.paused():
        new = _BatchMember.__new__
        ibytes = r['features']['image_bytes'].tolist()
        for b_, k in enumerate(batch):
            # (the per-graph array views are cut on first use: _BatchMember;
            # the attributes as one dictionary: a third of the time of ten
            # attribute stores)
            dg = out[k] = new(_BatchMember)
            dg.__dict__ = {
                '_b': r, '_k': b_, 'n_node': nl[b_], 'n_nz': nnz[b_],
                'weighted': weighted, 'node_t': node_t, 'edge_t': edge_t,
                'signature': signature, 'image_bytes': ibytes[b_],
                '_max_degree': maxdeg[b_]}
    return out


class _BatchMember(DeviceGraph):
    """One graph of a natively packed batch: its arrays are views into the
    batch's flat arrays, made when first asked for (most calls only need the
    sizes and the blob)."""

    def _nodes(self, key):
        a, z = self._b['node_off'][self._k], self._b['node_off'][self._k + 1]
        return self._b[key][a:z]

    def _nonzeros(self, key):
        a, z = self._b['nz_off'][self._k], self._b['nz_off'][self._k + 1]
        return self._b[key][a:z]

    perm = property(lambda self: self._nodes('perm'))
    rank = property(lambda self: self._nodes('rank'))
    degree = property(lambda self: self._nodes('degree'))
    adjacency_count = property(lambda self: self._nodes('count'))
    nz = property(lambda self: self._nonzeros('nz'))
    edge_index = property(lambda self: self._nonzeros('eid'))
    relocs = property(lambda self: self._b['no_relocs'])

    @property
    def offsets(self):
        try:
            return self._offsets
        except AttributeError:
            self._offsets = dict(zip(SECTIONS,
                                     self._b['sec_off_list'][self._k]))
            return self._offsets

    @property
    def rowptr(self):
        a, z = self._b['node_off'][self._k], self._b['node_off'][self._k + 1]
        return self._b['rowptr'][a + self._k:z + self._k + 1]

    @property
    def blob(self):
        o = self._b['blob_off']
        return self._b['blob'][o[self._k]:o[self._k + 1]]


def pack_many_numpy(graphs, real=np.float32):
    """Pack a whole list of graphs in one vectorised pass (numpy).

    Replaces one `DeviceGraph(graph)` call per graph (0.2-0.3 ms each, the
    dominant cost of a first kernel evaluation) by numpy operations over the
    concatenated node / edge tables of all graphs -- the batched counterpart
    of the reference's per-graph ``OctileGraph.__init__``
    (``graphdot/kernel/marginalized/_octilegraph.py:37-177``).  Every step
    restates the per-graph constructor above on flat arrays with a graph id
    per element, in an order that makes the results *byte-identical* to it
    (float32 degree sums in the same order, the same stable sorts; tested
    against the per-graph packer).  Graphs with variable-length attributes, or
    whose tables differ in columns / types from the first graph's, are packed
    one by one.  Returns the list of DeviceGraph objects; their blobs are
    views into one shared buffer.
    """
    real = np.dtype(real).type
    graphs = list(graphs)
    if not graphs:
        return []
    sig_n, sig_e = _scalar_frame(graphs[0].nodes), _scalar_frame(graphs[0].edges)
    batch = [k for k, g in enumerate(graphs)
             if sig_n is not None and sig_e is not None
             and _scalar_frame(g.nodes) == sig_n
             and _scalar_frame(g.edges) == sig_e
             and len(g.nodes._data['!i']) <= 0xFFFF
             and 2 * len(g.edges._data['!i']) <= 0xFFFF]
    out = [None] * len(graphs)
    for k in sorted(set(range(len(graphs))) - set(batch)):
        out[k] = DeviceGraph(graphs[k], real=real)
    if not batch:
        return out
    gs = [graphs[k] for k in batch]
    G = len(gs)

    def cat(frames, key):
        return np.concatenate([f._data[key] for f in frames])

    nframes, eframes = [g.nodes for g in gs], [g.edges for g in gs]
    n = np.array([len(f._data['!i']) for f in nframes], dtype=np.int64)
    m = np.array([len(f._data['!i']) for f in eframes], dtype=np.int64)
    node0 = np.concatenate(([0], np.cumsum(n)))          # first node of graph
    Nn, Ne = int(node0[-1]), int(m.sum())
    gid_n = np.repeat(np.arange(G), n)
    gid_e = np.repeat(np.arange(G), m)
    idx = cat(nframes, '!i').astype(np.int64)            # local node ids
    ei = cat(eframes, '!i').astype(np.int64)
    ej = cat(eframes, '!j').astype(np.int64)
    weighted = '!w' in eframes[0]
    w = (cat(eframes, '!w').astype(np.float32) if weighted
         else np.ones(Ne, np.float32))
    Ei, Ej = node0[gid_e] + ei, node0[gid_e] + ej        # global node ids

    # degrees: float32 sums in the per-graph order (ei pass, ej pass, loops)
    degree = np.zeros(Nn, np.float32)
    np.add.at(degree, Ei, w)
    np.add.at(degree, Ej, w)
    loops = ei == ej
    np.subtract.at(degree, Ei[loops], w[loops])
    degree[degree == 0] = 1.0

    # directed nonzeros: both orientations, duplicates collapse onto their
    # first occurrence (forward orientation first), sorted by (src, dst)
    src = np.concatenate((Ei, Ej))
    dst = np.concatenate((Ej, Ei))
    eid = np.concatenate((np.arange(Ne), np.arange(Ne)))
    key = src * (int(n.max()) + 1) + (dst - node0[np.concatenate((gid_e, gid_e))])
    order = np.argsort(key, kind='stable')
    keep = np.ones(len(order), dtype=bool)
    keep[1:] = key[order[1:]] != key[order[:-1]]
    first = order[keep]
    src, dst, eid = src[first], dst[first], eid[first]
    gid_z = np.concatenate((gid_e, gid_e))[first]

    # renumber the nodes of every graph by descending adjacency count (stable)
    count = np.bincount(src, minlength=Nn)
    local = np.arange(Nn) - node0[gid_n]
    perm_g = np.lexsort((local, -count, gid_n))          # new (global) -> old
    rank_g = np.empty(Nn, dtype=np.int64)
    rank_g[perm_g] = np.arange(Nn)                       # old -> new (global)
    rank_l = rank_g - node0[gid_n]                       # old -> new (local)
    perm_l = (perm_g - node0[gid_n]).astype(np.uint16)   # new -> old (local)
    degree_s = degree[perm_g]
    count_s = count[perm_g]
    src_n, dst_n = rank_g[src], rank_g[dst]              # global, new ids
    zorder = np.lexsort((dst_n, src_n))                  # CSR order, by graph
    src_n, dst_n, eid, gid_z = src_n[zorder], dst_n[zorder], eid[zorder], \
        gid_z[zorder]
    nnz = np.bincount(gid_z, minlength=G).astype(np.int64)
    nz0 = np.concatenate(([0], np.cumsum(nnz)))
    Nz = int(nz0[-1])
    nz_all = np.zeros(Nz, dtype=NZ_DTYPE)
    nz_all['i'] = src_n - node0[gid_z]
    nz_all['j'] = dst_n - node0[gid_z]
    # rowptr of graph g: [0, cumsum(count)] -> (n + 1) u16 entries per graph
    csum = np.cumsum(count_s)
    before = np.concatenate(([0], csum))[node0[:-1]]     # nonzeros before g
    rowptr_all = np.zeros(Nn + G, dtype=np.uint16)
    rp0 = node0[:-1] + np.arange(G)                      # start of g's rowptr
    rowptr_all[np.arange(Nn) + gid_n + 1] = csum - before[gid_n]

    # ---- record types from the first graph (phantom labels included) -------
    nodes0 = nframes[0].copy(deep=False)
    edges0 = eframes[0].copy(deep=False)
    if len(nodes0.columns) == 1:
        nodes0['labeled'] = np.zeros(len(nodes0), np.bool_)
    if len(edges0.columns) == 2:
        edges0['labeled'] = np.zeros(len(edges0), np.bool_)
    node_t = _widen(nodes0.drop(['!i']).rowtype(), real)
    label_t = _widen(edges0.drop(['!i', '!j', '!w']).rowtype(), real)
    edge_t = (np.dtype([('weight', real), ('label', label_t)], align=True)
              if weighted else label_t)
    nodes_aos = np.zeros(Nn, dtype=node_t)
    dest = node0[gid_n] + rank_l[node0[gid_n] + idx]     # row of node `idx`
    for name in node_t.names:
        if name == 'labeled' and name not in nframes[0]:
            continue
        nodes_aos[name][dest] = cat(nframes, name).astype(
            node_t.fields[name][0])
    edges_aos = np.zeros(Nz, dtype=edge_t)
    if weighted:
        edges_aos['weight'] = w[eid]
    if label_t.itemsize:
        target = edges_aos['label'] if weighted else edges_aos
        for name in label_t.names:
            if name == 'labeled' and name not in eframes[0]:
                continue
            target[name] = cat(eframes, name)[eid].astype(
                label_t.fields[name][0])
        if weighted:
            edges_aos['label'] = target

    # ---- blobs: one buffer, every section 16-byte aligned --------------------
    def pad(x):
        return (x + _ALIGN - 1) // _ALIGN * _ALIGN
    sizes = [4 * n, node_t.itemsize * n, 2 * (n + 1), 4 * nnz,
             edge_t.itemsize * nnz, 2 * n]
    offs, cursor = [], np.zeros(G, dtype=np.int64)
    for sz in sizes:
        offs.append(cursor.copy())
        cursor = cursor + pad(sz)
    blob_len = np.maximum(cursor, _ALIGN)
    blob0 = np.concatenate(([0], np.cumsum(blob_len)))
    buf = np.zeros(int(blob0[-1]), dtype=np.uint8)

    def scatter(records, rec_gid, rec_local, sec_off):
        """bytes of `records` (one per row) into buf at the section of their
        graph + local index * itemsize"""
        b = records.dtype.itemsize
        if b == 0 or len(records) == 0:
            return
        raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1, b)
        base = blob0[rec_gid] + sec_off[rec_gid] + rec_local * b
        buf[(base[:, None] + np.arange(b)[None, :]).ravel()] = raw.ravel()

    loc_n = np.arange(Nn) - node0[gid_n]
    loc_z = np.arange(Nz) - nz0[gid_z]
    scatter(degree_s, gid_n, loc_n, offs[0])
    scatter(nodes_aos, gid_n, loc_n, offs[1])
    gid_r = np.repeat(np.arange(G), n + 1)
    scatter(rowptr_all, gid_r, np.arange(Nn + G) - rp0[gid_r], offs[2])
    scatter(nz_all, gid_z, loc_z, offs[3])
    scatter(edges_aos, gid_z, loc_z, offs[4])
    scatter(perm_l, gid_n, loc_n, offs[5])

    signature = (weighted, str(node_t), str(edge_t))
    names = SECTIONS
    edge0 = np.concatenate(([0], np.cumsum(m)))
    maxdeg = np.maximum.reduceat(count_s, node0[:-1]).tolist() \
        if Nn else [0] * G
    offs_l = [o.tolist() for o in offs]
    for b_, k in enumerate(batch):
        dg = DeviceGraph.__new__(DeviceGraph)
        a, z = int(node0[b_]), int(node0[b_ + 1])
        za, zz = int(nz0[b_]), int(nz0[b_ + 1])
        dg.n_node, dg.n_nz, dg.weighted = int(n[b_]), int(nnz[b_]), weighted
        dg.perm = perm_l[a:z]
        dg.rank = rank_l[a:z]
        dg.degree = degree_s[a:z]
        dg.adjacency_count = count_s[a:z]
        dg.nz = nz_all[za:zz]
        dg.rowptr = rowptr_all[rp0[b_]:rp0[b_] + n[b_] + 1]
        dg.edge_index = eid[za:zz] - int(edge0[b_])
        dg.node_t, dg.edge_t, dg.signature = node_t, edge_t, signature
        dg.offsets = {name: offs_l[s_][b_] for s_, name in enumerate(names)}
        dg.image_bytes = _pad(dg.offsets['perm'] + 2 * dg.n_node)
        dg.relocs = np.zeros(0, dtype=np.int64)
        dg._max_degree = int(maxdeg[b_])
        dg.blob = buf[int(blob0[b_]):int(blob0[b_ + 1])]
        out[k] = dg
    return out


def class_bytes(n_node, n_nz):
    """Bytes of the label-class section that precedes a graph's blob in the
    arena: u8 node classes [pad4(n_node)] then u8 edge classes [pad4(n_nz)],
    padded to the section alignment.  mgk_solver.h computes the same."""
    n_node, n_nz = np.asarray(n_node), np.asarray(n_nz)
    return ((n_node + 3) // 4 * 4 + (n_nz + 3) // 4 * 4 + _ALIGN - 1) \
        // _ALIGN * _ALIGN


def _label_classes(dgraphs, vfields=None, efields=None, max_classes=255,
                   native=True):
    """Number the distinct node records and the distinct edge *labels* (the
    weight of a weighted edge is not part of its class) over all graphs,
    looking only at the attributes `vfields` / `efields` (None: all) -- the
    ones the microkernels read.  Returns None if some graph carries
    variable-length attributes or there are too many classes; else
    (node class ids of all graphs back to back, edge class ids likewise,
    representatives node_t[NV], representatives edge_t[NE])."""
    if not dgraphs or any(len(g.relocs) for g in dgraphs):
        return None
    g0 = dgraphs[0]
    node_t, edge_t = np.dtype(g0.node_t), np.dtype(g0.edge_t)

    def number(records, fields):
        dt = records.dtype
        if fields is not None:
            if not set(fields) <= set(dt.names or ()):
                return None
            keys = records[sorted(fields)] if fields else None
            if keys is not None:
                keys = repack_fields(keys)
        else:
            keys = records if dt.itemsize else None
        if keys is None or len(records) == 0 or keys.dtype.itemsize == 0:
            return np.zeros(len(records), np.int64), records[:1]
        if native:
            # gdh_number_records: the same numbering (np.unique order) from
            # the key byte ranges of the records, no repacked copy
            from ...hip import hostlib
            parts = [(0, dt.itemsize)] if fields is None else \
                [(dt.fields[f][1], dt.fields[f][0].itemsize)
                 for f in sorted(fields)]
            inv, first = hostlib.number_records(records, parts)
            rec = np.ascontiguousarray(records).view(np.uint8).reshape(
                len(records), dt.itemsize)[first]
            return inv.astype(np.int64), \
                np.ascontiguousarray(rec).view(dt).reshape(-1)
        raw = np.ascontiguousarray(keys).view(np.uint8).reshape(
            len(keys), keys.dtype.itemsize)
        if raw.shape[1] <= 8:
            # keys of up to 8 bytes are numbered as integers (a sort of
            # 64-bit words instead of a lexicographic sort of byte rows)
            wide = np.zeros((len(raw), 8), dtype=np.uint8)
            wide[:, :raw.shape[1]] = raw
            _, first, inv = np.unique(wide.view(np.uint64).reshape(-1),
                                      return_index=True, return_inverse=True)
        else:
            _, first, inv = np.unique(raw, axis=0, return_index=True,
                                      return_inverse=True)
        # representatives as byte rows: fancy indexing of a structured array
        # copies field by field and leaves its padding bytes undefined
        rec = np.ascontiguousarray(records).view(np.uint8).reshape(
            len(records), dt.itemsize)[first]
        return inv.reshape(-1), np.ascontiguousarray(rec).view(dt).reshape(-1)

    def gather(section, count, dt):
        """all records of one section of every graph, as one array"""
        if dt.itemsize == 0:
            return np.zeros(int(sum(count)), dt)
        b0 = whole_batch(dgraphs)
        if b0 is not None and native:
            from ...hip import hostlib
            return hostlib.gather_section(
                b0['blob'], b0['blob_off'], b0['sec_off'],
                SECTIONS.index(section), count, dt)
        if b0 is not None:
            # (plain lists of offsets into the batch's blob: no per-graph
            # views or offset dictionaries)
            blob, isz = b0['blob'], dt.itemsize
            col = SECTIONS.index(section)
            at = (np.asarray(b0['blob_off'][:-1], dtype=np.int64)
                  + b0['sec_off'][:, col]).tolist()
            raw = np.concatenate([blob[a:a + c * isz]
                                  for a, c in zip(at, count)])
        else:
            raw = np.concatenate([
                g.blob[g.offsets[section]:g.offsets[section] + c * dt.itemsize]
                for g, c in zip(dgraphs, count)])
        return raw.view(dt)

    feat = graph_features(dgraphs)
    nodes = gather('node', feat['n_node'].tolist(), node_t)
    edges = gather('edge', feat['n_nz'].tolist(), edge_t)
    numbered = number(nodes, vfields)
    if numbered is None:
        return None
    ncls, vrep = numbered
    if g0.weighted:
        label_t = edge_t.fields['label'][0]
        numbered = number(np.ascontiguousarray(edges['label']), efields)
        if numbered is None:
            return None
        ecls, lrep = numbered
        erep = np.zeros(max(len(lrep), 1), dtype=edge_t)
        erep['weight'] = 1
        if label_t.itemsize and len(lrep):
            erep['label'] = lrep
    else:
        numbered = number(edges, efields)
        if numbered is None:
            return None
        ecls, erep = numbered
        if len(erep) == 0:
            erep = np.zeros(1, dtype=edge_t)
    if len(vrep) == 0:
        vrep = np.zeros(1, dtype=node_t)
    if len(vrep) > max_classes or len(erep) > max_classes:
        return None
    # (flat: the node classes of all graphs back to back, likewise the edges)
    return ncls.astype(np.uint8), ecls.astype(np.uint8), vrep, erep


class GraphArena:
    """Concatenation of DeviceGraph blobs + the graph_t header table, ready
    for one host-to-device copy.  Layout::

        [headers][node class representatives][edge class representatives]
        [classes of graph 0][blob 0][classes of graph 1][blob 1]...

    The class section of a graph (`class_bytes`) sits directly in front of
    its blob so that the solver stages [classes | degree .. perm] into LDS
    with one contiguous copy.  `classes` is None when the labels cannot be
    numbered (`_label_classes`); the sections are then zero."""

    def __init__(self, dgraphs, vfields=None, efields=None, classes=True,
                 native=True):
        self.n = len(dgraphs)
        hdr_bytes = _pad(self.n * HEADER_DTYPE.itemsize)
        cls = _label_classes(dgraphs, vfields, efields, native=native) \
            if classes else None
        self.classes = None
        cursor = hdr_bytes
        if cls is not None:
            ncls, ecls, vrep, erep = cls
            self.classes = dict(nv=len(vrep), ne=len(erep), vrep=cursor,
                                erep=cursor + _pad(vrep.nbytes))
            cursor = self.classes['erep'] + _pad(erep.nbytes)
        feat = graph_features(dgraphs) if self.n else None
        b0 = whole_batch(dgraphs)
        sizes = np.diff(np.asarray(b0['blob_off'], dtype=np.int64)) \
            if b0 is not None else np.array([len(g.blob) for g in dgraphs],
                                            dtype=np.int64)
        cbytes = class_bytes(feat['n_node'], feat['n_nz']) \
            if self.n else np.zeros(0, np.int64)
        ends = cursor + np.cumsum(sizes + cbytes)
        starts = ends - sizes if self.n else np.zeros(0, np.int64)
        self.nbytes = int(ends[-1]) if self.n else int(cursor)
        self.host = np.zeros(self.nbytes, dtype=np.uint8)
        self.blob_start = starts
        self.class_bytes = cbytes
        if cls is not None:
            c = self.classes
            self.host[c['vrep']:c['vrep'] + vrep.nbytes] = \
                vrep.view(np.uint8).ravel() if vrep.nbytes else []
            self.host[c['erep']:c['erep'] + erep.nbytes] = \
                erep.view(np.uint8).ravel() if erep.nbytes else []
        self._relocs = []
        hdr = np.zeros(self.n, dtype=HEADER_DTYPE)
        natively = native and b0 is not None and self.n > 0
        if natively:
            # blobs and class sections in one native pass (gdh_assemble_arena)
            from ...hip import hostlib
            hostlib.assemble_arena(
                b0['blob'], b0['blob_off'], starts, cbytes, feat['n_node'],
                feat['n_nz'], None if cls is None else ncls,
                None if cls is None else ecls, self.host)
        elif b0 is not None and not cbytes.any():
            # the batch's blobs are back to back already: one copy
            self.host[cursor:] = b0['blob']
        elif b0 is not None:
            host, blob, bo = self.host, b0['blob'], b0['blob_off']
            for k, s in enumerate(starts.tolist()):
                host[s:s + bo[k + 1] - bo[k]] = blob[bo[k]:bo[k + 1]]
        else:
            for g, s in zip(dgraphs, starts.tolist()):
                self.host[s:s + len(g.blob)] = g.blob
                if len(g.relocs):
                    self._relocs.append(g.relocs + s)
                    words = self.host
                    for where in g.relocs + s:
                        word = words[where:where + 8].view(np.uint64)
                        word[0] += np.uint64(s)
        if self.n:
            n_node, n_nz = feat['n_node'], feat['n_nz']
            hdr['n_node'], hdr['n_nz'] = n_node, n_nz
            for s_, name in enumerate(SECTIONS):   # arena-relative for now
                hdr[name] = starts + (
                    b0['sec_off'][:, s_] if b0 is not None else np.array(
                        [g.offsets[name] for g in dgraphs], dtype=np.int64))
            hdr['hist'] = feat['hist']
            if cls is not None and not natively:
                # class ids in front of every blob: [node classes, padded to
                # 4][edge classes], one scatter per kind
                c0 = starts - cbytes
                node0 = np.cumsum(n_node) - n_node
                self.host[np.repeat(c0 - node0, n_node)
                          + np.arange(int(n_node.sum()))] = ncls
                c1 = c0 + (n_node + 3) // 4 * 4
                nz0 = np.cumsum(n_nz) - n_nz
                self.host[np.repeat(c1 - nz0, n_nz)
                          + np.arange(int(n_nz.sum()))] = ecls
        self._hdr = hdr
        self._relocs = (np.concatenate(self._relocs) if self._relocs
                        else np.zeros(0, np.int64))
        self.n_node = hdr['n_node'].astype(np.int64)
        self.n_nz = hdr['n_nz'].astype(np.int64)

    def relocated(self, base):
        """Byte image with every pointer rebased onto device address `base`."""
        if self.nbytes >= 2**32:
            raise ValueError('graph arena exceeds the 4 GiB offset range')
        img = self.host.copy()
        img[:self._hdr.nbytes] = self._hdr.view(np.uint8)
        for where in self._relocs:
            word = img[where:where + 8].view(np.uint64)
            word[0] += np.uint64(base)
        return img
