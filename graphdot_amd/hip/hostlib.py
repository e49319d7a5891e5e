"""ctypes binding of libgdhost.so (include/gdhost.h, csrc/gdhost.cpp): the
native host side of the path -- graph packer, label-class numbering, solver
variant classification, job layout.  Host code only: g++ builds it, no HIP,
no device; it is compiled on first use (and by ``__graft_entry__.build()``)
and a failure to build or load raises -- there is no silent numpy path in the
product (the numpy implementations stay as the specification the tests hold
the native results to).
"""
import ctypes
import os
import subprocess
import threading
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), 'csrc')
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(_HERE)), 'include')
LIB_PATH = os.path.join(CSRC, 'libgdhost.so')
_lock = threading.Lock()
_lib = None

_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
SIGNATURES = {
    'gdh_pack_graphs': [_i32] + [_vp] * 7 + [_i32, _vp, _i32, _i32, _i32, _i32,
                                            _vp, _i64] + [_vp] * 10
    + [_i64, _vp, _vp],
    'gdh_number_records': [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp],
    'gdh_classify_oc': [_i64] + [_vp] * 7 + [_i32] + [_vp] * 6
    + [_i32, _i32, _i64, _i32, _vp, _vp, _vp],
    'gdh_pair_keys': [_vp, _i64, _vp, _i32, _i32, _vp, _vp],
    'gdh_order_jobs': [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp],
    'gdh_pairwise_jobs': [_i64, _i64, _vp],
    'gdh_gather_section': [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _i64],
    'gdh_assemble_arena': [_i64] + [_vp] * 9 + [_i64],
}


class HostLibError(RuntimeError):
    pass


def build_library(force=False):
    """g++ csrc/gdhost.cpp -> csrc/libgdhost.so."""
    src = os.path.join(CSRC, 'gdhost.cpp')
    hdr = os.path.join(INCLUDE, 'gdhost.h')
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src),
                                                  os.path.getmtime(hdr))):
        return LIB_PATH
    cxx = os.environ.get('CXX', 'g++')
    tmp = LIB_PATH + f'.{os.getpid()}.tmp'
    cmd = [cxx, '-O2', '-fPIC', '-shared', '-std=c++17', '-pthread',
           f'-I{INCLUDE}', src,
           '-o', tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise HostLibError(f'building libgdhost.so failed:\n{r.stderr}')
    os.replace(tmp, LIB_PATH)          # (ranks of a multi-GPU run may race)
    return LIB_PATH


def lib():
    global _lib
    with _lock:
        if _lib is None:
            L = ctypes.CDLL(build_library())
            for name, argtypes in SIGNATURES.items():
                fn = getattr(L, name)
                fn.argtypes = argtypes
                fn.restype = ctypes.c_int
            L.gdh_version.restype = ctypes.c_char_p
            L.gdh_version.argtypes = []
            _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _check(rc, what):
    if rc != 0:
        raise HostLibError(f'{what} failed with code {rc} '
                           '(-1: bad argument, -2: capacity, -3: exception inside '
                           'the library, e.g. out of memory)')


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def pack_graphs(node_off, edge_off, node_id, ei, ej, w, node_rec, label_rec,
                edge_size, label_offset, weight_bytes):
    """gdh_pack_graphs on numpy arrays; returns a dict of the outputs (flat
    arrays trimmed to their used length)."""
    G = len(node_off) - 1
    node_off, edge_off = _c(node_off, np.int64), _c(edge_off, np.int64)
    Nn, Ne = int(node_off[-1]), int(edge_off[-1])
    node_id, ei, ej = (_c(a, np.int64) for a in (node_id, ei, ej))
    w = None if w is None else _c(w, np.float32)
    node_size = node_rec.dtype.itemsize
    label_size = label_rec.dtype.itemsize if label_rec is not None else 0
    node_raw = np.ascontiguousarray(node_rec).view(np.uint8) \
        if node_size else np.zeros(0, np.uint8)
    label_raw = np.ascontiguousarray(label_rec).view(np.uint8) \
        if label_size else np.zeros(0, np.uint8)
    n = np.diff(node_off)
    m = np.diff(edge_off)

    def pad(x):
        return (x + 15) // 16 * 16
    cap = int((pad(4 * n) + pad(node_size * n) + pad(2 * (n + 1))
               + pad(8 * m) + pad(edge_size * 2 * m) + pad(2 * n)).sum()
              + 16 * G)
    zcap = 2 * Ne
    out = dict(
        blob=np.empty(max(cap, 16), np.uint8),
        blob_off=np.zeros(G + 1, np.int64), sec_off=np.zeros((G, 6), np.int64),
        nnz=np.zeros(G, np.int64), perm=np.zeros(Nn, np.uint16),
        rank=np.zeros(Nn, np.int64), degree=np.zeros(Nn, np.float32),
        count=np.zeros(Nn, np.int64), rowptr=np.zeros(Nn + G, np.uint16),
        nz=np.zeros(2 * max(zcap, 1), np.uint16),
        eid=np.zeros(max(zcap, 1), np.int64), nz_off=np.zeros(G + 1, np.int64),
        maxdeg=np.zeros(G, np.int64))
    _check(lib().gdh_pack_graphs(
        G, _p(node_off), _p(edge_off), _p(node_id), _p(ei), _p(ej), _p(w),
        _p(node_raw), node_size, _p(label_raw), label_size, int(edge_size),
        int(label_offset), int(weight_bytes), _p(out['blob']),
        out['blob'].nbytes, _p(out['blob_off']), _p(out['sec_off']),
        _p(out['nnz']), _p(out['perm']), _p(out['rank']), _p(out['degree']),
        _p(out['count']), _p(out['rowptr']), _p(out['nz']), _p(out['eid']),
        zcap, _p(out['nz_off']), _p(out['maxdeg'])), 'gdh_pack_graphs')
    Nz = int(out['nz_off'][-1])
    out['blob'] = out['blob'][:int(out['blob_off'][-1])]
    out['nz'] = out['nz'][:2 * Nz]
    out['eid'] = out['eid'][:Nz]
    return out


def number_records(records, parts):
    """Class id of every record and the index of the first record of every
    class; `parts`: (byte offset, length) ranges of the key inside a record,
    in key order."""
    records = np.ascontiguousarray(records)
    n, size = len(records), records.dtype.itemsize
    raw = records.view(np.uint8) if size else np.zeros(0, np.uint8)
    off = np.array([p[0] for p in parts], dtype=np.int32)
    ln = np.array([p[1] for p in parts], dtype=np.int32)
    cls = np.zeros(n, np.int32)
    first = np.zeros(max(n, 1), np.int64)
    nc = ctypes.c_int64(0)
    _check(lib().gdh_number_records(
        _p(raw), n, size, _p(off), _p(ln), len(parts), _p(cls), _p(first),
        ctypes.cast(ctypes.byref(nc), ctypes.c_void_p)),
        'gdh_number_records')
    return cls, first[:nc.value]


def classify_oc(ca, cb, n_node, n_nz, image_bytes, maxdeg, hist, variants, C,
                real_size, lds_limit, fly_min_degree=8, extra_lds=None):
    """Variant index (into `variants`, a list of (W, S, R, D, L or None)) per
    class pair, or -1; and NP per pair."""
    ca, cb = _c(ca, np.int32), _c(cb, np.int32)
    nv = len(variants)
    W = np.array([v[0] for v in variants], np.int32)
    S = np.array([v[1] for v in variants], np.int32)
    R = np.array([v[2] for v in variants], np.int32)
    D = np.array([v[3] for v in variants], np.int32)
    nL = np.array([len(v[4]) if v[4] else 0 for v in variants], np.int32)
    L = np.zeros((max(nv, 1), 12), np.int32)
    for k, v in enumerate(variants):
        if v[4]:
            if len(v[4]) > 12:
                raise ValueError('static layouts have at most 12 batches')
            L[k, :len(v[4])] = v[4]
    choice = np.zeros(len(ca), np.int32)
    NP = np.zeros(len(ca), np.int64)
    extra = np.zeros(max(nv, 1), np.int64)
    if extra_lds is not None:
        extra[:nv] = extra_lds
    _check(lib().gdh_classify_oc(
        len(ca), _p(ca), _p(cb), _p(_c(n_node, np.int32)),
        _p(_c(n_nz, np.int32)), _p(_c(image_bytes, np.int64)),
        _p(_c(maxdeg, np.int32)), _p(_c(hist, np.uint16)), nv, _p(W), _p(S),
        _p(R), _p(D), _p(nL), _p(L), int(C), int(real_size), int(lds_limit),
        int(fly_min_degree), _p(extra), _p(choice), _p(NP)),
        'gdh_classify_oc')
    return choice, NP


def pair_keys(jobs, cid, nc):
    """(pk per job, job count per key) for the class ids `cid` of the
    graphs."""
    jobs = np.ascontiguousarray(jobs)
    raw = jobs.view(np.uint32)
    cid = _c(cid, np.int32)
    pk = np.empty(len(jobs), np.int32)
    count = np.zeros(nc * nc, np.int64)
    _check(lib().gdh_pair_keys(_p(raw), len(jobs), _p(cid), len(cid), int(nc),
                               _p(pk), _p(count)), 'gdh_pair_keys')
    return pk, count


def order_jobs(pk, rank_of_key, n_ranks, jobs=None):
    """Job ids in launch order: stable counting sort by rank_of_key[pk].
    With `jobs` also returns the job records in that order."""
    pk = _c(pk, np.int32)
    rank_of_key = _c(rank_of_key, np.int32)
    order = np.empty(len(pk), np.uint32)
    raw = out = None
    if jobs is not None:
        jobs = np.ascontiguousarray(jobs)
        raw = jobs.view(np.uint32)
        out = np.empty_like(jobs)
    _check(lib().gdh_order_jobs(
        _p(pk), len(pk), _p(rank_of_key), len(rank_of_key), int(n_ranks),
        _p(order), _p(raw), None if out is None else _p(out.view(np.uint32))),
        'gdh_order_jobs')
    return order if jobs is None else (order, out)


def pairwise_jobs(nx, ny=None, dtype=None):
    """(i, j) records of a kernel-matrix evaluation: the upper triangle with
    the diagonal (ny None), or all pairs (i, nx + j)."""
    n = nx * (nx + 1) // 2 if ny is None else nx * ny
    jobs = np.empty(2 * n, np.uint32)
    _check(lib().gdh_pairwise_jobs(int(nx), -1 if ny is None else int(ny),
                                   _p(jobs)), 'gdh_pairwise_jobs')
    return jobs if dtype is None else jobs.view(dtype)


def gather_section(blob, blob_off, sec_off, col, count, dtype):
    """Records of section `col` of every graph of a packed batch as one
    array of `dtype`."""
    dtype = np.dtype(dtype)
    count = _c(count, np.int64)
    blob_off = _c(blob_off, np.int64)
    out = np.empty(int(count.sum()), dtype=dtype)
    raw = out.view(np.uint8) if dtype.itemsize else out
    _check(lib().gdh_gather_section(
        _p(blob), _p(blob_off), _p(_c(sec_off, np.int64)), int(col), _p(count),
        len(count), dtype.itemsize, _p(raw), out.nbytes), 'gdh_gather_section')
    return out


def assemble_arena(blob, blob_off, starts, cbytes, n_node, n_nz, ncls, ecls,
                   host):
    """Blobs (and label-class sections) of a packed batch into the arena
    image `host` (uint8)."""
    _check(lib().gdh_assemble_arena(
        len(starts), _p(blob), _p(_c(blob_off, np.int64)),
        _p(_c(starts, np.int64)), _p(_c(cbytes, np.int64)),
        _p(_c(n_node, np.int64)), _p(_c(n_nz, np.int64)),
        None if ncls is None else _p(_c(ncls, np.uint8)),
        None if ecls is None else _p(_c(ecls, np.uint8)),
        _p(host), host.nbytes), 'gdh_assemble_arena')


# --------------------------------------------------------------------------
# CPython helper: the attribute tables of a list of graphs as flat columns
# --------------------------------------------------------------------------
_collect = None


def collector():
    """The `_gdcollect` extension module (csrc/gdcollect.cpp, built with g++
    against this interpreter's headers on first use), or False when it cannot
    be built -- no Python.h on the machine: `pack_many` then gathers the
    tables in Python, the path the tests hold the native one to."""
    global _collect
    if _collect is None:
        with _lock:
            if _collect is None:
                _collect = _build_collector()
    return _collect


def _build_collector():
    import importlib.machinery
    import importlib.util
    import sysconfig
    src = os.path.join(CSRC, 'gdcollect.cpp')
    suffix = sysconfig.get_config_var('EXT_SUFFIX') or '.so'
    out = os.path.join(CSRC, '_gdcollect' + suffix)
    inc = sysconfig.get_paths().get('include')
    try:
        if not (os.path.exists(out)
                and os.path.getmtime(out) >= os.path.getmtime(src)):
            if not inc or not os.path.exists(os.path.join(inc, 'Python.h')):
                return False
            tmp = out + f'.{os.getpid()}.tmp'
            r = subprocess.run(
                [os.environ.get('CXX', 'g++'), '-O2', '-fPIC', '-shared',
                 '-std=c++17', f'-I{inc}', f'-I{np.get_include()}', src,
                 '-o', tmp],
                capture_output=True, text=True)
            if r.returncode != 0:
                return False
            os.replace(tmp, out)
        spec = importlib.util.spec_from_file_location(
            '_gdcollect', out,
            loader=importlib.machinery.ExtensionFileLoader('_gdcollect', out))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except Exception:
        return False


def same_tables(graphs):
    """True if the node and the edge tables of all `graphs` have the first
    graph's columns and element types (one native pass, csrc/gdcollect.cpp
    `same_tables`); None if not, or not decidable natively (no extension,
    object columns): the caller compares row types in Python."""
    mod = collector()
    if not mod or not hasattr(mod, 'same_tables'):
        return None
    graphs = graphs if isinstance(graphs, list) else list(graphs)
    try:
        return (mod.same_tables(graphs, 'nodes')
                and mod.same_tables(graphs, 'edges')) or None
    except (AttributeError, TypeError):
        return None


def collect_columns(graphs, attr, keys, dtypes):
    """{key: column of all graphs back to back}, rows per graph (int64) --
    or None if the tables are not uniform (or there is no extension).
    `dtypes`: {key: numpy dtype of the first graph's column}."""
    mod = collector()
    if not mod:
        return None
    res = mod.collect(graphs, attr, tuple(keys))
    if res is None:
        return None
    cols, lengths, formats = res
    out = {}
    for k, raw, (_, isz) in zip(keys, cols, formats):
        dt = np.dtype(dtypes[k])
        if dt.itemsize != isz or dt.hasobject:
            return None
        out[k] = np.frombuffer(raw, dtype=dt)
    return out, np.frombuffer(lengths, dtype=np.int64)
