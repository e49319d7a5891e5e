"""HIP backend of the marginalized graph kernel for MI355X (gfx950).

Serves the reference's backend seam (``Backend.__call__``,
``graphdot/kernel/marginalized/_backend_cuda.py:247-368``): graph images are
packed and cached per graph, node/edge microkernels are printed as HIP C++
functors and JIT-compiled with hipcc, hyperparameters are handed to the
kernel as by-value arguments, and the job list is partitioned over a small
menu of register-resident solver variants (see ``csrc/device/mgk_solver.h``).

There is no CPU path: a missing ``libgdhip.so`` / hipcc / device raises.
"""
import copy
import os
import re
import uuid
import zlib
from collections import OrderedDict, namedtuple
import numpy as np
from ...codegen import Template
from ...util import gcpause
from ...util.cookie import IdentityCache
from ...codegen.sympy_printer import to_real_expr
from ...codegen.typetool import _dtype_util
from ...hip import jit, runtime
from ...microkernel import TensorProduct, Product
from ...util.iterable import flatten, fold_like
from ._backend import Backend
from ._devicegraph import (DeviceGraph, GraphArena, HIST_BINS, class_bytes,
                           degree_histograms, graph_features, pack_many)

_TEMPLATE = os.path.join(os.path.dirname(__file__), 'template.hip')

# solver flags, mirror graphdot::mgk::F_* (mgk_solver.h)
F_NODAL, F_DIAGONAL, F_SYMMETRIC, F_LMIN1, F_BLOCK, F_PACKED, F_REFCOMPAT, \
    F_DENSE = 1, 2, 4, 8, 16, 32, 64, 128

Variant = namedtuple('Variant', 'W S R')
#: owner-computes solver (csrc/device/mgk_oc.h): S nonzero slots and R rows
#: per lane, for pairs of graphs whose largest degree is at most D
#: L: None, or the static row-batch layout (seg_layout<L...> in mgk_oc.h):
#: batch k of a lane owns exactly L[k] slots, S = sum(L), R = len(L)
OCVariant = namedtuple('OCVariant', 'W S R D L', defaults=(None,))


#: measured efficiency of M workgroups on one pair of the streamed solver
#: against M times one workgroup (MI355X, protein-like graphs of 150-600
#: atoms, float; profiles/sessions.md r5_session18 and DESIGN.md section 4c:
#: three grid-wide barriers per iteration, B staged by every part)
STREAM_PART_EFFICIENCY = ((1, 1.0), (2, 0.74), (4, 0.58), (8, 0.53),
                          (16, 0.50), (32, 0.49), (64, 0.46), (128, 0.35),
                          (256, 0.26))


def stream_parts(costs, count, resident, queued):
    """Workgroups per pair (M) of a streamed launch: the M with the smallest
    estimated time for the launch's pairs -- `costs`, largest first, in
    arbitrary units.  One workgroup per pair (`queued` workgroups, the
    hardware hands the next pair to the first free compute unit) takes
    max(largest pair, sum / resident); M workgroups per pair form
    resident // M groups that walk the pairs round-robin, a group's time is
    the sum of its pairs over M x efficiency(M).  136 pairs of 16 graphs:
    M = 1 105.8 ms (the largest pair on one workgroup, 120 compute units
    idle), 2 / 4 / 8 / 16: 84.8 / 73.7 / 65.8 / 65.7 ms; 528 pairs: 140.9 ms
    at M = 1 against 191-198 at 2 ... 16."""
    if costs is None or len(costs) == 0:
        return max(1, min(resident // max(count, 1), STREAM_MAX_PARTS))
    c = np.asarray(costs, dtype=np.float64)
    best, best_t = 1, max(c[0], c.sum() / max(1, min(resident, queued)))
    table = dict(STREAM_PART_EFFICIENCY)
    ms, es = zip(*STREAM_PART_EFFICIENCY)
    # (powers of two, and what gives every pair a group of its own)
    own = min(resident // max(len(c), 1), STREAM_MAX_PARTS)
    for M in sorted(set(ms[1:]) | ({own} if own > 1 else set())):
        if M > min(resident, STREAM_MAX_PARTS):
            break
        eff = table.get(M) or float(np.interp(np.log2(M), np.log2(ms), es))
        groups = max(1, resident // M)
        pad = -len(c) % groups
        load = np.concatenate((c, np.zeros(pad))).reshape(-1, groups).sum(0)
        t = load.max() / (M * eff)
        if t < 0.95 * best_t:        # (a larger M must earn its barriers)
            best, best_t = M, t
    return best


def OCStatic(*L, D=4):
    """One-wave owner-computes variant with the static layout L."""
    return OCVariant(1, int(sum(L)), len(L), D, tuple(int(x) for x in L))


#: register-resident solver menu, cheapest first.  A pair fits a variant if
#: its stage-1 walk needs <= S slots per lane and N = n1*n2 <= 64*W*R.
VARIANTS = [
    Variant(1, 8, 2), Variant(1, 12, 3), Variant(1, 16, 4), Variant(1, 20, 5),
    Variant(1, 24, 6), Variant(1, 28, 7), Variant(1, 32, 9), Variant(1, 48, 9),
    Variant(1, 64, 16),
    Variant(4, 16, 4), Variant(4, 32, 8), Variant(4, 64, 16),
    Variant(16, 16, 2), Variant(16, 32, 4), Variant(16, 64, 8),
    Variant(16, 128, 16),
]
#: owner-computes menu, cheapest first: molecular graphs (every degree <= 4)
#: take the D = 4 kernels, graphs with degrees up to 8 (the Newman-Watts-
#: Strogatz graphs of configuration 2: 4..7) the D = 8 kernels with 1, 4, 8
#: or 16 waves per pair.  S <= 64: larger slot arrays are not promoted to
#: registers by the compiler (they would live in scratch memory).
#: Static layouts first: a pair whose per-batch degree products fit one takes
#: it (no flush tests, row sums in registers); the dynamic variants behind
#: them take what is left.  The layouts are the profiles of molecular graphs
#: (degrees <= 4: the first batch holds the 16-term rows of two four-valent
#: atoms, then 4 = 2 x 2 / 4 x 1, then hydrogens): on the QM7-like set 66
#: distinct profiles, all under one of these.
OC_STATIC_VARIANTS = [
    OCStatic(16), OCStatic(16, 4), OCStatic(16, 4, 1), OCStatic(16, 4, 4),
    OCStatic(16, 4, 4, 1), OCStatic(16, 4, 4, 1, 1), OCStatic(16, 4, 4, 3, 1),
    OCStatic(16, 4, 4, 3, 1, 1), OCStatic(16, 4, 4, 4, 1, 1, 1),
    OCStatic(16, 4, 4, 4, 3, 1, 1, 1), OCStatic(16, 4, 4, 4, 4, 1, 1, 1, 1),
]
#: on-the-fly variants (mgk_oc.h FLY: S = 0, D = 0): no register slots, the
#: edge microkernel is evaluated per term in every iteration; any degree.
#: For the pairs no slot variant fits (dense from_ase-like graphs): values
#: (graph-level and nodal) and graph-level value + gradient.
OC_FLY_VARIANTS = [
    OCVariant(4, 0, 1, 0), OCVariant(4, 0, 2, 0), OCVariant(4, 0, 3, 0),
    OCVariant(8, 0, 2, 0), OCVariant(16, 0, 2, 0), OCVariant(16, 0, 4, 0),
]
OC_VARIANTS = OC_STATIC_VARIANTS + [
    OCVariant(1, 12, 2, 4), OCVariant(1, 16, 3, 4), OCVariant(1, 20, 3, 4),
    OCVariant(1, 20, 4, 4), OCVariant(1, 24, 4, 4), OCVariant(1, 28, 5, 4),
    OCVariant(1, 28, 6, 4), OCVariant(1, 32, 7, 4), OCVariant(1, 36, 9, 4),
    OCVariant(1, 32, 3, 8), OCVariant(1, 48, 5, 8), OCVariant(1, 64, 9, 8),
    OCVariant(4, 32, 3, 8), OCVariant(4, 40, 2, 8), OCVariant(4, 48, 4, 8),
    OCVariant(4, 64, 5, 8),
    OCVariant(8, 40, 2, 8), OCVariant(8, 48, 3, 8), OCVariant(8, 64, 4, 8),
    OCVariant(16, 32, 2, 8), OCVariant(16, 40, 2, 8), OCVariant(16, 48, 3, 8),
    OCVariant(16, 64, 3, 8),
] + OC_FLY_VARIANTS
#: pairs with a node of more than this many neighbours are what the
#: on-the-fly variants are for (the slot variants stop at degree 8)
FLY_MIN_DEGREE = 8
#: sentinel: the global-memory general solver (any pair size)
GENERAL = Variant(0, 0, 0)
#: sentinel: the kernel that fills the global microkernel tables
TABLES = Variant(-1, 0, 0)
GENERAL_THREADS = 1024
#: sentinel: the dense-tile solver on the matrix cores (csrc/device/mgk_mfma.h):
#: float value solves of DENSE pairs of graphs of at most 32 nodes under an
#: edge microkernel that ignores the labels (`Constant`: one separable term).
#: 82 M pairs/s against 4.2 M of the on-the-fly dense product on the dense
#: molecular set (scripts/mfma_experiment.py).  GD_MFMA=0: off.
MFMA = Variant(-3, 0, 0)
MFMA_MAX_NODES = 32
#: sentinel: the streamed solver for large pairs (csrc/device/mgk_stream.h):
#: value solves of pairs beyond every register- and LDS-resident variant; one
#: graph of the pair staged in LDS, the other streamed row by row, CG vectors
#: in a global scratch.
STREAM = Variant(-2, 0, 0)
STREAM_THREADS = 1024
#: rows of p staged per pass and row group (mgk_stream.h A_ROWS)
STREAM_ROWS = {4: 8, 8: 8}       # by the size of a real
#: neighbours per lane segment of the LDS-resident graph (mgk_stream.h SEG_CAP)
STREAM_CAP = 16
#: workgroups that may share one pair of the streamed solver
STREAM_MAX_PARTS = 256
#: dynamic LDS a pair of the streamed solver may ask for (mgk_stream.h LDS_BUDGET)
STREAM_LDS_BUDGET = 159 * 1024
LDS_LIMIT = 160 * 1024
_LARGE_PAIR_SOLVERS = [MFMA, STREAM, GENERAL]
#: independent pairs (waves) per workgroup of the one-wave variants (mgk_solver.h
#: WPB; four pairs per 256-thread workgroup were 4-5 % slower: a workgroup's
#: LDS and wave slots are only released when its slowest pair is done)
WPB1 = 1
#: microkernel value tables (label classes) are used when they fit this much
#: LDS per workgroup: (n_node_classes^2 + n_edge_classes^2) reals
TABLE_LDS_LIMIT = 8 * 1024
#: class pairs (node + edge) up to which the global tables are used
GLOBAL_TABLE_LIMIT = 1 << 16


def _real_name(real):
    return {np.dtype(np.float32): 'float', np.dtype(np.float64): 'double'}[
        np.dtype(real)]


# --------------------------------------------------------------------------
# struct printing and packing
# --------------------------------------------------------------------------
def declstruct(dtype, name):
    """C++ definition ``struct name {...};`` of an aligned numpy struct dtype.
    Unlike ``decltype`` every nested struct is *named* (``name_field``) so
    that zero-size members can be declared ``constexpr static`` (not allowed
    inside unnamed structs in standard C++)."""
    dtype = np.dtype(dtype)
    members = []
    for key in dtype.names or ():
        ft = dtype.fields[key][0]
        if key.startswith('$'):
            n, tpl, *args = key[1:].split('::')
            targs = ','.join(np.dtype(a).name for a in args)
            members.append(f'{tpl}<{targs}> {n};')
        elif _dtype_util.is_object(ft):
            if ft.itemsize == 0:
                members.append(f'constexpr static _empty {key} {{}};')
            else:
                sub = f'{name}_{key}'
                members.append(declstruct(ft, sub)[:-1] + f' {key};')
        elif _dtype_util.is_array(ft):
            dims = ''.join(f'[{d}]' for d in ft.shape)
            members.append(f'{ft.base.name} {key}{dims};')
        else:
            members.append(f'{ft.name} {key};')
    return f'struct {name} {{' + ' '.join(members) + '};'


def packable(dtype):
    """Can two records of this struct dtype ride in one record of pairs
    (`declstruct2`)?  Every leaf a 4-byte number, no arrays, no
    variable-length attributes, no padding."""
    dtype = np.dtype(dtype)
    if dtype.names is None:
        return False

    def leaves(t):
        n = 0
        for key in t.names:
            ft = t.fields[key][0]
            if key.startswith('$') or _dtype_util.is_array(ft):
                return -1
            if _dtype_util.is_object(ft):
                if ft.itemsize == 0:
                    continue
                sub = leaves(ft)
                if sub < 0:
                    return -1
                n += sub
            elif ft.kind in 'fiu' and ft.itemsize == 4:
                n += 1
            else:
                return -1
        return n
    n = leaves(dtype)
    return n > 0 and 4 * n == dtype.itemsize


def declstruct2(dtype, name):
    """`declstruct` with every 4-byte leaf a pair, ``graphdot::pk2<T>``: two
    records side by side, leaf by leaf (device/fmath.h; mgk_oc.h evaluates the
    edge microkernel of the dense product on two terms at once)."""
    dtype = np.dtype(dtype)
    members = []
    for key in dtype.names:
        ft = dtype.fields[key][0]
        if _dtype_util.is_object(ft):
            if ft.itemsize == 0:
                members.append(f'constexpr static _empty {key} {{}};')
            else:
                members.append(declstruct2(ft, f'{name}_{key}')[:-1]
                               + f' {key};')
        else:
            members.append(f'graphdot::pk2<{ft.name}> {key};')
    return f'struct {name} {{' + ' '.join(members) + '};'


#: calls a packed evaluation can make (overloaded for pairs in device/fmath.h)
_PACKED_CALLS = re.compile(
    r'graphdot::(?:exp|ipow<\d+>|ripow<\d+>)$')


def packed_expression(expr):
    """Does the generated expression only call what device/fmath.h overloads
    for pairs?"""
    calls = re.findall(r'([A-Za-z_][\w:]*(?:<[^<>()]*>)?)\s*\(', expr)
    return all(_PACKED_CALLS.match(c) for c in calls)


def widen_theta(dtype, real):
    """theta structs hold float32 hyperparameters in the reference; an fp64
    build stores them as float64."""
    dtype = np.dtype(dtype)
    if np.dtype(real) == np.float32:
        return dtype
    if dtype.names is not None:
        return np.dtype([(k, widen_theta(dtype.fields[k][0], real))
                         for k in dtype.names], align=True)
    if dtype.subdtype is not None:
        return np.dtype((widen_theta(dtype.base, real), dtype.shape))
    return np.dtype(real) if dtype == np.float32 else dtype


def raw_state(obj):
    """Like ``obj.state`` but without the float32 rounding of cpptype, so an
    fp64 build sees the hyperparameters at full precision."""
    values = getattr(obj, '_theta_values', None)
    if values is not None:
        return tuple(values.values())
    out = []
    dt = obj.dtype
    for key in dt.names or ():
        ft = dt.fields[key][0]
        v = getattr(obj, key)
        out.append(raw_state(v) if _dtype_util.is_object(ft) else v)
    return tuple(out)


def fill_struct(view, dtype, state):
    """Write nested tuple `state` into the 0-d structured array `view`."""
    for key, value in zip(dtype.names or (), state):
        ft = dtype.fields[key][0]
        if _dtype_util.is_object(ft):
            if ft.itemsize:
                fill_struct(view[key], ft, value)
        else:
            view[key] = value


def pack_theta(obj, real):
    dt = widen_theta(obj.dtype, real)
    buf = np.zeros((), dtype=dt) if dt.itemsize else None
    if buf is not None:
        fill_struct(buf, dt, raw_state(obj) if np.dtype(real) != np.float32
                    else obj.state)
    return dt, buf


# --------------------------------------------------------------------------
# backend
# --------------------------------------------------------------------------
class ClassPairs:
    """The jobs of a call by pair of graph classes: `pk[t]` the key of job t,
    `upk` the sorted keys in use (position = index into the per-class-pair
    results), `count[key]` the jobs per key."""

    def __init__(self, pk, upk, count, nc):
        self.pk, self.upk, self.count, self.nc = pk, upk, count, nc
        self._sel = None

    @property
    def sel(self):
        """index of every job's class pair in the per-class-pair results"""
        if self._sel is None:
            pos = np.zeros(self.nc * self.nc, dtype=np.int32)
            pos[self.upk] = np.arange(len(self.upk), dtype=np.int32)
            self._sel = pos[self.pk]
        return self._sel

    @property
    def members(self):
        return self.count[self.upk].astype(np.int64)


class Partition(tuple):
    """Result of `HIPBackend._partition`: (jobs, used, order_all, launches)
    as a tuple, plus `.jobs_sorted` -- the job records in launch order when
    the native ordering wrote them (else None) -- and `.merge_map`, the
    launch merging that was applied ({variant index: variant index})."""
    jobs_sorted = None
    merge_map = {}


class NotOwnerComputes(Exception):
    """Some pair of the call does not fit an owner-computes solver variant
    (raised when a feature only those solvers have was asked for)."""


class Plan:
    """Everything resident on the device for one kernel evaluation."""
    pass


class Layout:
    """The hyperparameter-independent part of a Plan (HIPBackend._layout)."""
    pass


#: code objects loaded in this process, by JIT cache key (shared by backends)
_MODULES = {}
#: rendered translation units, by (code signature, variant, build options):
#: pure text work on hyperparameter-independent inputs, shared by backends
_SOURCES = {}
#: HIP streams / events are expensive to create (milliseconds each): every
#: LaunchSet draws from this process-wide pool, in order
_STREAM_POOL, _EVENT_POOL = [], []
_LOW_STREAM_POOL, _LOW_EVENT_POOL = [], []     # streams of the lowest priority


class _DeviceGraphList(list):
    """The packed graphs of one call, with the tuple of their identities
    (the key of the arena and layout caches) computed once."""
    ids = None


def _ids(dgraphs):
    return getattr(dgraphs, 'ids', None) or tuple(map(id, dgraphs))


class LaunchSet:
    """The streams and events that run the launches of a plan, step after
    step, without a host synchronisation: the launches every solver depends
    on (the table kernel) go to the null stream, every solver variant to a
    non-blocking stream of its own (longest first), ordered device-side --
    the solver streams wait for the `start` event of the step (recorded on
    the null stream behind the table kernel and behind everything the caller
    enqueued there before, e.g. the previous step's collective), the null
    stream waits for every solver stream.  `events[k] = (begin, end)` are
    recorded around solver launch k on the stream it runs on."""

    def __init__(self):
        self.streams, self.done = [], []
        # detached launches (`enqueue(detached=True)`) run on streams of the
        # lowest priority: what overlaps them on the null stream -- a chain of
        # small dependent launches, the Cholesky factorisation -- gets compute
        # units as soon as it has a workgroup to dispatch instead of queueing
        # behind the solver grids
        self.low_streams, self.low_done = [], []
        self._last_done = self.done
        self.start = runtime.Event()
        self._n_last = 0
        self.max_streams = int(os.environ.get('GD_MAX_STREAMS', '3'))
        self.streams_forced = 'GD_MAX_STREAMS' in os.environ

    def enqueue(self, plan, events=None, serial=False, front=None, after=(),
                detached=False):
        """`front`: a stream that takes the place of the null stream for the
        table kernel and the start event (pipelined steps: the null stream
        may still be busy with the previous step's collective).  It first
        waits for the events in `after` and for the solver launches of the
        previous `enqueue` (they read the table buffer this step rewrites).
        `detached`: the null stream does NOT wait for the solver streams --
        what the caller enqueues there next (the Gaussian process factors the
        kernel matrix while the gradient solves run) overlaps the solvers;
        `join()` makes it wait later."""
        fh = None
        if front is not None and not serial:
            fh = front.h
            for e in after:
                front.wait_event(e)
            for slot in range(self._n_last):
                front.wait_event(self._last_done[slot])
        for L in plan.pre_launches:
            runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                           stream=fh, dynamic_lds=L['dynamic_lds'],
                           cooperative=L.get('cooperative', False))
        n = len(plan.launches)
        if serial:
            for k, L in enumerate(plan.launches):
                if events is not None:
                    events[k][0].record()
                runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                               dynamic_lds=L['dynamic_lds'],
                           cooperative=L.get('cooperative', False))
                if events is not None:
                    events[k][1].record()
            self._n_last = 0
            return
        # longest launch first, each to the stream with the least work so far
        # (at most `max_streams` streams: the runtime multiplexes them onto a
        # few hardware queues, and every stream costs two cross-queue
        # dependencies per step)
        cost = [plan.launches[k]['count'] * (plan.launches[k]['variant'].S + 8)
                for k in range(n)]
        order = sorted(range(n), key=lambda k: -cost[k])
        ns = max(1, min(n, self.max_streams or n))
        # (a plan may ask for fewer: HIPBackend.prepare, `stream_hint`)
        hint = getattr(plan, 'stream_hint', None)
        if hint and not self.streams_forced:
            ns = max(1, min(ns, hint))
        low = detached
        streams, done, spool, epool = (
            (self.low_streams, self.low_done, _LOW_STREAM_POOL,
             _LOW_EVENT_POOL) if low else
            (self.streams, self.done, _STREAM_POOL, _EVENT_POOL))
        while len(streams) < ns:
            k = len(streams)
            if len(spool) <= k:
                spool.append(runtime.Stream(low_priority=low))
                epool.append(runtime.Event())
            streams.append(spool[k])
            done.append(epool[k])
        self.start.record(fh)
        load = [0] * ns
        for slot in range(ns):
            streams[slot].wait_event(self.start)
        for k in order:
            slot = load.index(min(load))
            load[slot] += cost[k]
            L, s = plan.launches[k], streams[slot]
            if events is not None:
                events[k][0].record(s.h)
            runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                           stream=s.h, dynamic_lds=L['dynamic_lds'],
                           cooperative=L.get('cooperative', False))
            if events is not None:
                events[k][1].record(s.h)
        for slot in range(ns):
            done[slot].record(streams[slot].h)
            if not detached:
                runtime.null_stream_wait_event(done[slot])
        self._n_last, self._last_done = ns, done
        self._detached = ns if detached else 0

    def join(self):
        """The null stream waits for the solver launches of a detached
        `enqueue` (device-side)."""
        for slot in range(getattr(self, '_detached', 0)):
            runtime.null_stream_wait_event(self._last_done[slot])
        self._detached = 0


class HIPBackend(Backend):
    """MI355X backend.

    Parameters
    ----------
    device: int or None
        HIP device ordinal (default: ``$LOCAL_RANK`` or 0).
    real: numpy float32 (reference arithmetic, default) or float64
    jobs_per_unit: int
        Graph pairs handled by one wave (small pairs) or workgroup (large
        pairs) before it retires; sets the launch grid.  Default 1.
    hipcc_extra: list of str
        Extra compiler flags (also ``$GD_HIPCC_EXTRA``).
    variants: list of Variant
        Solver menu ``(W, S, R)`` in order of preference; ``GENERAL`` (the
        global-scratch solver for pairs of any size) last.
    record_iterations: bool
        Keep per-job CG iteration counts on the device (`iterations(plan)`).
    occupancy: dict (W, S) -> waves per SIMD, or None
        Overrides the measured occupancy targets (also ``$GD_OCCUPANCY``).
    concurrent: bool
        One HIP stream per solver variant (default) or all on one stream.
    tables: 'global' (default), 'lds' or False
        Microkernel values per pair of label classes from tables instead of
        per nonzero pair (see __init__).
    min_launch: int
        Owner-computes launches of fewer waves (pairs x waves per pair) are
        merged into the next larger compatible variant in use (default 8192;
        0: never).
    nodal_gradient_in_kernel: bool
        Nodal Jacobians inside the launch (default) or by re-launches.
    native: bool
        Host side of a call in C++ (libgdhost.so, default) or in numpy.
    """

    @staticmethod
    def array(ndarray):
        return np.array(ndarray, copy=True)

    @staticmethod
    def zeros(size, dtype=np.float32):
        return np.zeros(size, dtype)

    @staticmethod
    def empty(size, dtype=np.float32):
        """Output buffers of the caller (reference: managed memory,
        graphdot/cuda/array.py:14-31): large ones come from a pool of pinned
        host blocks, so that `collect` copies the result by DMA straight into
        the array the user gets (hip/runtime.py::pinned_empty)."""
        return runtime.pinned_empty(size, dtype)

    def __init__(self, **kwargs):
        self.uuid = uuid.uuid4()
        self.device = kwargs.pop('device', None)
        self.real = np.dtype(kwargs.pop('real', np.float32)).type
        self.jobs_per_unit = int(kwargs.pop(
            'jobs_per_unit', 1))
        self.hipcc_extra = list(kwargs.pop('hipcc_extra', [])) + \
            os.environ.get('GD_HIPCC_EXTRA', '').split()
        self.variants = list(kwargs.pop(
            'variants', OC_VARIANTS + VARIANTS + _LARGE_PAIR_SOLVERS))
        if os.environ.get('GD_VARIANTS'):  # experiments: "W:S:R[:D],L16x4x1,..."
            def parse(item):
                if item.startswith('L'):
                    return OCStatic(*map(int, item[1:].split('x')))
                f = list(map(int, item.split(':')))
                return OCVariant(*f) if len(f) == 4 else Variant(*f)
            self.variants = [parse(item) for item in
                             os.environ['GD_VARIANTS'].split(',')] \
                + _LARGE_PAIR_SOLVERS
        self.record_iterations = kwargs.pop('record_iterations', False)
        self.occupancy = kwargs.pop('occupancy', None)
        self.concurrent = kwargs.pop('concurrent', True)
        # microkernel value tables over label classes (GraphArena.classes):
        #   'global' (default): one tiny launch per evaluation fills a global
        #       table per pair of classes, the owner-computes solvers look
        #       values up (mgk_oc.h) -- whenever the labels can be numbered;
        #   'lds' (or True): the two-stage solvers rebuild the tables per
        #       workgroup in LDS (round 1; slower than direct evaluation with
        #       one pair per workgroup, kept for microkernels much more
        #       expensive than a Gaussian); the owner-computes menu is off;
        #   False: every microkernel value is evaluated where it is used.
        tables = kwargs.pop('tables', 'global')
        if tables is True:
            tables = 'lds'
        if tables not in (False, 'lds', 'global'):
            raise ValueError(f'tables={tables!r}: False, "lds" or "global"')
        self.tables = tables
        self._launch_set = None
        self.min_launch = int(kwargs.pop(
            'min_launch', os.environ.get('GD_MIN_LAUNCH', 8192)))
        self.nodal_gradient_in_kernel = bool(kwargs.pop(
            'nodal_gradient_in_kernel', True))
        # host side of a call (graph packing, label classes, variant per job,
        # launch order): libgdhost.so (csrc/gdhost.cpp), or -- native=False --
        # the numpy restatements it is tested against
        self.native = bool(kwargs.pop(
            'native', True))
        if self.native:
            from ...hip import hostlib
            hostlib.lib()                  # fail loudly if it cannot be built
            hostlib.collector()            # (optional helper: loaded here, not in the first call)
        if self.occupancy is None and os.environ.get('GD_OCCUPANCY'):
            # e.g. GD_OCCUPANCY="1:16:5,1:24:4"  (W:S:waves)
            self.occupancy = {
                (int(a), int(b)): int(c) for a, b, c in
                (item.split(':') for item in
                 os.environ['GD_OCCUPANCY'].split(','))}
        if kwargs:
            raise TypeError(f'unknown HIPBackend options {sorted(kwargs)}')
        runtime.lib()                      # fail loudly if the library is absent
        self._modules = _MODULES           # (tu key) -> runtime.Module
        self._arenas = OrderedDict()       # tuple(id(DeviceGraph)) -> (arena, buf)
        self._pool = {}                    # name -> DeviceBuffer (grow-only)
        self._dgraph_lists = IdentityCache()
        self._layouts = OrderedDict()      # job-list key -> Layout (LRU)
        self._source_cache = _SOURCES      # (code signature, variant) -> text
        self._module_sets = {}             # (code signature, variants) -> (modules, argument dtype)
        self.layout_cache_size = 8
        self._props = None
        self.last_plan = None

    def __deepcopy__(self, memo):
        # clones of a kernel (clone_with_theta) share device state, like the
        # reference backend (_backend_cuda.py:63-64)
        return copy.copy(self)

    # -- device helpers -------------------------------------------------------
    @property
    def props(self):
        if self._props is None:
            self._props = runtime.device_props(self.device)
        return self._props

    @staticmethod
    def _zeroed_buffer(nbytes):
        """A device buffer of zeros (the barrier cells of the streamed
        solver; owned by the plan that asked for it)."""
        buf = runtime.DeviceBuffer(max(int(nbytes), 256))
        buf.upload(np.zeros(buf.nbytes, dtype=np.uint8))
        runtime.synchronize()
        return buf

    def _buffer(self, name, nbytes):
        """Grow-only pool of per-call device buffers.  A buffer that is too
        small is *dropped from the pool*, never freed here: earlier plans and
        the zero-copy views handed out by `device_gram` (torch tensors alias
        them) hold references, and `DeviceBuffer.__del__` releases the
        allocation when the last of those is gone."""
        buf = self._pool.get(name)
        if buf is None or buf.nbytes < nbytes:
            buf = self._pool[name] = runtime.DeviceBuffer(
                max(int(nbytes * 1.25), 256))
        return buf

    # -- graphs ---------------------------------------------------------------
    def _register_graph(self, graph):
        key = (self.uuid.int, np.dtype(self.real).str)   # (an int hashes in C)
        if key not in graph.cookie:
            graph.cookie[key] = DeviceGraph(graph, real=self.real)
        return graph.cookie[key]

    @staticmethod
    def _assert_homogeneous(x, y):
        if (x.weighted != y.weighted or x.node_t != y.node_t
                or x.edge_t != y.edge_t):
            raise TypeError(
                'All nodes/edges must be of the same type: '
                f'{x.node_t} / {x.edge_t} vs {y.node_t} / {y.edge_t}. '
                'If the graph attributes match in name but differ in type, '
                'try to normalize automatically with '
                '`Graph.unify_datatype`.')

    @staticmethod
    def _used_fields(kernel):
        """Attributes a microkernel reads, taken from the code it generates:
        ``x1.<name>`` / ``x2.<name>`` accesses.  () for a kernel that ignores
        its arguments (Constant), None (= every attribute) if an argument is
        used as a whole."""
        fun, jac = kernel.gen_expr('x1', 'x2')
        text = ' '.join([fun, *jac])
        names = set(re.findall(r'\bx[12]\.([A-Za-z_]\w*)', text))
        bare = re.search(r'\bx[12]\b(?!\.[A-Za-z_])', text)
        return None if bare else tuple(sorted(names))

    def _table_bytes(self, arena):
        """LDS bytes of the per-workgroup microkernel tables (tables='lds')
        of `arena`'s label classes, or 0 if this call evaluates the
        microkernels directly (other modes, labels not numberable, or tables
        beyond TABLE_LDS_LIMIT)."""
        c = arena.classes
        if self.tables != 'lds' or c is None:
            return 0
        b = (c['nv']**2 + c['ne']**2) * np.dtype(self.real).itemsize
        return int(-(-b // 16) * 16) if b <= TABLE_LDS_LIMIT else 0

    def _global_tables(self, arena):
        """Do the owner-computes solvers of this call read the microkernels
        from the per-evaluation global tables (tables='global' and the labels
        fall into classes)?"""
        c = arena.classes
        return (self.tables == 'global' and c is not None
                and c['nv']**2 + c['ne']**2 <= GLOBAL_TABLE_LIMIT)

    def _host_arena(self, dgraphs, fields):
        # (label classes are only numbered when the tables are in use)
        return GraphArena(dgraphs, *fields, classes=bool(self.tables),
                          native=self.native)

    def _arena(self, dgraphs, fields=(None, None)):
        key = (_ids(dgraphs), fields if self.tables else None)
        hit = self._arenas.get(key)
        if hit is not None:
            self._arenas.move_to_end(key)
            return hit
        arena = self._host_arena(dgraphs, fields)
        buf = runtime.DeviceBuffer(arena.nbytes)
        buf.upload(arena.relocated(buf.ptr))
        runtime.synchronize()
        self._arenas[key] = (arena, buf, list(dgraphs))
        # evicted arenas are released when the last layout / plan that points
        # into them is gone (DeviceBuffer frees on destruction)
        while len(self._arenas) > 4:
            self._arenas.popitem(last=False)
        return self._arenas[key]

    # -- code generation --------------------------------------------------------
    @staticmethod
    def gencode_kernel(kernel, name, real=np.float32):
        """HIP C++ for a microkernel: ``struct <name>_theta_t`` (the packed
        hyperparameters) and functor ``<name>_t`` with ``operator()`` and
        ``_j_a_c_o_b_i_a_n_`` (reference: _backend_cuda.py:157-193)."""
        fun, jac = kernel.gen_expr('x1', 'x2')
        rn = _real_name(real)
        fun = to_real_expr(fun, np.dtype(real).name)
        jac = [to_real_expr(j, np.dtype(real).name) for j in jac]
        return Template(r'''
${theta_t}
struct ${name}_t : ${name}_theta_t {
    constexpr static int jac_dims = ${jac_dims};
    template<class X> __device__ __forceinline__
    auto operator() (X const &x1, X const &x2) const {
        return ${expr};
    }
    template<class X> __device__ __forceinline__
    auto _j_a_c_o_b_i_a_n_(X const &x1, X const &x2) const {
        graphdot::array<real_t, jac_dims> j;
        ${jac;}
        return j;
    }
};
''').render(
            name=name, jac_dims=len(jac), expr=fun,
            theta_t=declstruct(widen_theta(kernel.dtype, real),
                               f'{name}_theta_t'),
            jac=[f'j[{i}] = {e};' for i, e in enumerate(jac)] + [''],
        ).replace('real_t', 'real_t') + f'// real = {rn}\n'

    @staticmethod
    def gencode_probability(pfunc, name, real=np.float32):
        """Functor for the starting probability on a node ``n``
        (reference: _backend_cuda.py:195-228)."""
        fun, jac = pfunc.gen_expr()
        fun = to_real_expr(fun, np.dtype(real).name)
        jac = [to_real_expr(j, np.dtype(real).name) for j in jac]
        return Template(r'''
${theta_t}
struct ${name}_t : ${name}_theta_t {
    constexpr static int jac_dims = ${jac_dims};
    template<class N> __device__ __forceinline__
    auto operator() (N const &n) const {
        return ${expr};
    }
    template<class N> __device__ __forceinline__
    auto _j_a_c_o_b_i_a_n_(N const &n) const {
        graphdot::array<real_t, jac_dims> j;
        ${jac;}
        return j;
    }
};
''').render(
            name=name, jac_dims=len(jac), expr=fun,
            theta_t=declstruct(widen_theta(pfunc.dtype, real),
                               f'{name}_theta_t'),
            jac=[f'j[{i}] = {e};' for i, e in enumerate(jac)] + [''],
        )

    @staticmethod
    def pack_state(obj, diff_grid=False, diff_eps=1e-2):
        """[state] or, with diff_grid, [state, state(theta_0 e^+eps),
        state(theta_0 e^-eps), ...] (reference: _backend_cuda.py:230-245)."""
        pack = [obj.state]
        if diff_grid is True:
            logtheta = np.log(list(flatten(obj.theta)))
            for i in range(len(logtheta)):
                for delta in (diff_eps, -diff_eps):
                    o = copy.deepcopy(obj)
                    t = logtheta.copy()
                    t[i] += delta
                    o.theta = fold_like(np.exp(t), o.theta)
                    pack.append(o.state)
        return pack

    def _params_dtype(self, node_kernel, edge_kernel, p):
        def theta(obj):
            dt = widen_theta(obj.dtype, self.real)
            return dt if dt.itemsize else np.dtype(np.uint8)
        P = np.uintp
        return np.dtype([
            ('arena', P), ('jobs', P), ('order', P), ('starts', P),
            ('gramian', P), ('gradient', P), ('iters', P), ('scratch', P),
            ('sync', P), ('tables', P), ('diag', P), ('diag_grad', P), ('hotspot', P),
            ('node_starts', P), ('diag_ld', np.uint32),
            ('n_launch_jobs', np.uint32), ('nX', np.uint32),
            ('nY', np.uint32), ('nJ', np.uint32), ('flags', np.uint32),
            ('order_offset', np.uint32), ('u_capacity', np.uint32),
            ('g_capacity', np.uint32), ('parts', np.uint32),
            ('n_vclass', np.uint32), ('n_eclass', np.uint32),
            ('vrep', np.uint32), ('erep', np.uint32),
            ('q', self.real), ('q0', self.real), ('eps', self.real),
            ('ftol', self.real), ('gtol', self.real),
            ('node_kernel', theta(node_kernel)),
            ('edge_kernel', theta(edge_kernel)),
            ('p_start', theta(p)),
        ], align=True)

    def _params_fd_dtype(self, node_kernel, edge_kernel, p):
        """Kernel arguments of the nodal finite-difference gradient solver:
        graphdot::mgk::params_fd_t (mgk_oc.h)."""
        def theta(obj):
            dt = widen_theta(obj.dtype, self.real)
            return dt if dt.itemsize else np.dtype(np.uint8)
        nv = len(node_kernel.gen_expr('x1', 'x2')[1])
        ne = len(edge_kernel.gen_expr('x1', 'x2')[1])
        return np.dtype([
            ('base', self._params_dtype(node_kernel, edge_kernel, p)),
            ('q_plus', self.real), ('q_minus', self.real),
            ('node_theta', self.real, (max(nv, 1),)),
            ('edge_theta', self.real, (max(ne, 1),)),
            ('node_diff', theta(node_kernel), (max(2 * nv, 1),)),
            ('edge_diff', theta(edge_kernel), (max(2 * ne, 1),)),
        ], align=True)

    def kernel_name(self, v, C, nodal=False, tab=False, ngrad=False,
                    maximin=False):
        """Entry point name: arithmetic, solver variant, flavour."""
        f = 'f64' if np.dtype(self.real) == np.float64 else 'f32'
        if v == GENERAL:
            return f'mgk_{f}_general_T{GENERAL_THREADS}_C{C}'
        if v == TABLES:
            return f'mgk_{f}_tables_C{C}'
        if v == MFMA:
            return f'mgk_{f}_mfma_C{C}'
        if v == STREAM:
            return f'mgk_{f}_stream_T{STREAM_THREADS}_C{C}'
        if isinstance(v, OCVariant):
            return f'mgk_{f}_oc{v.D}_W{v.W}_S{v.S}_R{v.R}_C{C}' + \
                ('_L' + 'x'.join(map(str, v.L)) if v.L else '') + \
                ('_nodal' if nodal else '') + ('_tab' if tab else '') + \
                ('_ngrad' if ngrad else '') + ('_maximin' if maximin else '')
        return f'mgk_{f}_W{v.W}_S{v.S}_R{v.R}_C{C}' + \
            ('_nodal' if nodal else '') + ('_tab' if tab else '')

    #: occupancy targets of the fp32 value solver, W = 1 (hipcc 7.2, gfx950):
    #: S -> waves per SIMD.  Spill-free or nearly so, except S = 24 where 16
    #: VGPRs spilled around (never inside) the CG loop buy 4 instead of 3
    #: waves: +7 % on the whole Gram matrix for 2 KB of scratch traffic per
    #: pair.  S = 16 at 6 waves and S = 28 at 4 are another +3 % but write
    #: 5-10 KB of scratch per pair to HBM (profiles/README.md) -- not taken.
    _WAVES_F32_VALUE = {8: 6, 12: 6, 16: 5, 20: 4, 24: 4, 28: 3, 32: 2}
    #: same for the value + gradient solver (C = 2) and the fp64 value solver:
    #: the fastest of a per-variant sweep (scripts/occupancy_sweep2.sh)
    _WAVES_F32_GRADIENT = {8: 5, 12: 4, 16: 3, 20: 2, 24: 2, 28: 2, 32: 1}
    _WAVES_F64_VALUE = {8: 5, 12: 4, 16: 3, 20: 3, 24: 2, 28: 2, 32: 2}
    _WAVES_F64_GRADIENT = {8: 3, 12: 3, 16: 2, 20: 2, 24: 1, 28: 1, 32: 1}
    _WAVES_F32_VALUE_W4 = {(4, 32): 3, (4, 64): 2}
    _WAVES_F64_VALUE_W4 = {(4, 32): 2}

    def waves_per_eu(self, v, C):
        """Occupancy target handed to the register allocator
        (amdgpu_waves_per_eu).  The fp32 value solver needs about
        5.3 S + 4 R + 6 VGPRs to stay spill-free (measured: S=12 -> 81,
        S=16 -> 106, S=24 -> 150, S=32 -> 211); the targets are the highest
        occupancies with at most a handful of spilled registers, which were
        also the fastest or within 2 % of it in a per-variant sweep
        (scripts/occupancy_sweep.sh) and keep scratch traffic out of HBM."""
        floor = -(-64 * v.W * (WPB1 if v.W == 1 else 1) // 256)  # block must fit
        if self.occupancy is not None and (v.W, v.S) in self.occupancy:
            return max(self.occupancy[(v.W, v.S)], floor)
        if isinstance(v, OCVariant):
            return max(self._oc_waves(v, C), -(-64 * v.W // 256))
        if v.W == 1:
            f64 = np.dtype(self.real) == np.float64
            table = {(1, False): self._WAVES_F32_VALUE,
                     (2, False): self._WAVES_F32_GRADIENT,
                     (1, True): self._WAVES_F64_VALUE,
                     (2, True): self._WAVES_F64_GRADIENT}.get((C, f64), {})
            if v.S in table:
                return max(table[v.S], floor)
        # four-wave variants of the fp32 value solver, measured on the
        # 8..48-node random graphs of configuration 2
        if C == 1:
            w4 = self._WAVES_F32_VALUE_W4 \
                if np.dtype(self.real) == np.float32 \
                else self._WAVES_F64_VALUE_W4
            if (v.W, v.S) in w4:
                return max(w4[(v.W, v.S)], floor)
        need = 5.3 * v.S + 4 * v.R + 6
        if C == 2:
            need = 1.6 * need
        if np.dtype(self.real) == np.float64:   # every real takes two VGPRs
            need = 1.8 * need
        for n in (8, 6, 5, 4, 3, 2):
            if need <= (512 // n) // 8 * 8 + 4:
                return max(n, floor)
        return max(1, floor)

    #: occupancy targets of the owner-computes variants, the fastest of a
    #: per-variant sweep on MI355X (scripts/oc_sweep.py: QM7-like set for
    #: D = 4, configuration 2 for D = 8): (double?, C) -> {(W, S, R, D): waves}
    #: (float (28, 5) at 5 waves is 2 % faster than at 4 but writes 2 KB of
    #: scratch per pair to HBM, profiles/r02_f32_pmc.csv: not taken)
    _OC_WAVES = {
        (False, 1): {('L', 16): 6, ('L', 16, 4): 6, ('L', 16, 4, 1): 5, ('L', 16, 4, 4): 5,
                     ('L', 16, 4, 4, 1): 3, ('L', 16, 4, 4, 1, 1): 3,
                     ('L', 16, 4, 4, 3, 1): 4, ('L', 16, 4, 4, 3, 1, 1): 4,
                     ('L', 16, 4, 4, 4, 1, 1, 1): 3,
                     ('L', 16, 4, 4, 4, 3, 1, 1, 1): 3,
                     ('L', 16, 4, 4, 4, 4, 1, 1, 1, 1): 3,
                     (1, 12, 2, 4): 6, (1, 16, 3, 4): 6, (1, 20, 3, 4): 6,
                     (1, 20, 4, 4): 4, (1, 24, 4, 4): 5, (1, 28, 5, 4): 4,
                     (1, 28, 6, 4): 4, (1, 32, 7, 4): 4, (1, 36, 9, 4): 2,
                     (1, 32, 3, 8): 2, (1, 48, 5, 8): 2, (1, 64, 9, 8): 2,
                     (4, 32, 3, 8): 4, (4, 40, 2, 8): 4, (4, 48, 4, 8): 2,
                     (4, 64, 5, 8): 3, (8, 40, 2, 8): 4, (8, 48, 3, 8): 4,
                     (8, 64, 4, 8): 4, (16, 32, 2, 8): 4, (16, 40, 2, 8): 4,
                     (16, 48, 3, 8): 4, (16, 64, 3, 8): 4},
        (True, 1): {('L', 16): 4, ('L', 16, 4): 4, ('L', 16, 4, 1): 3, ('L', 16, 4, 4): 3,
                    # (round 5: the five-batch layouts at three waves -- with
                    # the row sums zeroed late and p updated in place
                    # (mgk_oc.h) their iteration is spill-free at 168
                    # registers: (16,4,4,1,1) 0.864 -> 0.726 ms, (16,4,4,3,1)
                    # 0.115 -> 0.105; the six-batch layout reloads 12 values
                    # per iteration at three -- 0.557 -> 0.849 -- unless its
                    # diagonals leave the registers, below)
                    ('L', 16, 4, 4, 1): 3, ('L', 16, 4, 4, 1, 1): 3,
                    # ((16,4,4,3,1,1): three waves with the Jacobi diagonals in
                    # LDS, mgk_oc.h DLDS -- nothing reloaded in the iteration)
                    ('L', 16, 4, 4, 3, 1): 3, ('L', 16, 4, 4, 3, 1, 1): 3,
                    ('L', 16, 4, 4, 4, 1, 1, 1): 2,
                    ('L', 16, 4, 4, 4, 3, 1, 1, 1): 2,
                    ('L', 16, 4, 4, 4, 4, 1, 1, 1, 1): 2,
                    (1, 12, 2, 4): 4, (1, 16, 3, 4): 4, (1, 20, 3, 4): 4,
                    (1, 20, 4, 4): 3, (1, 24, 4, 4): 3, (1, 28, 5, 4): 3,
                    (1, 28, 6, 4): 3, (1, 32, 7, 4): 2, (1, 36, 9, 4): 2,
                    (1, 64, 9, 8): 1, (4, 32, 3, 8): 3, (4, 40, 2, 8): 3,
                    (4, 64, 5, 8): 2, (8, 40, 2, 8): 2, (8, 64, 4, 8): 2},
        (False, 2): {('L', 16): 4, ('L', 16, 4): 4, ('L', 16, 4, 1): 4, ('L', 16, 4, 4): 2,
                     ('L', 16, 4, 4, 1): 2, ('L', 16, 4, 4, 1, 1): 3,
                     ('L', 16, 4, 4, 3, 1): 3, ('L', 16, 4, 4, 3, 1, 1): 3,
                     ('L', 16, 4, 4, 4, 1, 1, 1): 2,
                     ('L', 16, 4, 4, 4, 3, 1, 1, 1): 2,
                     ('L', 16, 4, 4, 4, 4, 1, 1, 1, 1): 2,
                     (1, 12, 2, 4): 4, (1, 16, 3, 4): 4, (1, 20, 3, 4): 3,
                     (1, 20, 4, 4): 2, (1, 24, 4, 4): 3, (1, 28, 5, 4): 3,
                     (1, 28, 6, 4): 2, (1, 32, 7, 4): 2, (1, 36, 9, 4): 2},
        # (static layouts: the sequential solves of mgk_oc.h SEQ, round 4 --
        # profiles/sessions.md r4_session2 / r4_session4: three waves only
        # where the loop stays free of scratch reloads, the three-batch kernel)
        # (round 5: the four-batch sequential solver at three waves -- with
        # the row sums zeroed late and p updated in place nothing is reloaded
        # inside its iteration at 168 registers: 2.89 -> 2.65 ms; (16,4,4)
        # loses at three, 0.175 -> 0.188, the five-batch kernel reloads three
        # values per iteration there, 2.0 -> 2.86 ms)
        (True, 2): {('L', 16): 2, ('L', 16, 4): 2, ('L', 16, 4, 1): 3, ('L', 16, 4, 4): 2,
                    ('L', 16, 4, 4, 1): 3, ('L', 16, 4, 4, 1, 1): 2,
                    ('L', 16, 4, 4, 3, 1): 2, ('L', 16, 4, 4, 3, 1, 1): 2,
                    ('L', 16, 4, 4, 4, 1, 1, 1): 2,
                    ('L', 16, 4, 4, 4, 3, 1, 1, 1): 2,
                    ('L', 16, 4, 4, 4, 4, 1, 1, 1, 1): 1,
                    (1, 12, 2, 4): 3, (1, 16, 3, 4): 3, (1, 20, 3, 4): 3,
                    (1, 20, 4, 4): 2, (1, 24, 4, 4): 2, (1, 28, 5, 4): 2,
                    (1, 28, 6, 4): 2, (1, 32, 7, 4): 2, (1, 36, 9, 4): 2},
    }

    #: static layouts that lose to the dynamic variants behind them in the
    #: menu (measured, scripts/oc_sweep.py): (double?, C) -> {layout, ...}.
    #: Round 3 had the double value + gradient solves here: the STACKED
    #: two-right-hand-side solver moves 16 bytes per gathered element and
    #: keeps both systems' vectors in registers, and its static form was no
    #: faster than the dynamic one (14.3 / 18.4 / 22.7 ns per pair for the
    #: three- / four- / five-batch pairs against 14.5 / 17.2 / 21.7).  The
    #: static double kernels now solve the two systems one after the other
    #: (mgk_oc.h SEQ) and are back in the menu.
    _STATIC_OFF = {}

    def _static_enabled(self, v, C):
        f64 = np.dtype(self.real) == np.float64
        return v.L not in self._STATIC_OFF.get((f64, C), ())

    _FLY_WAVES = 3     # (occupancy 3-6 waves per SIMD: 4.95-4.86 M pairs/s, round 3)

    def _oc_waves(self, v, C, ngrad=False):
        """Occupancy target of an owner-computes variant: per lane S values +
        S gather indices, 6 registers per row (x, r, p, diagonal, its inverse,
        the publish address), the gathers in flight and ~24 others; a double
        takes two registers."""
        f64 = np.dtype(self.real) == np.float64
        if v.S == 0:
            # on-the-fly kernels: no slot arrays, but the unrolled term loops
            # keep ~140 registers busy (spill-free at three waves per SIMD)
            return max(self._FLY_WAVES, -(-64 * v.W // 256))
        # (static layouts under ('L',) + layout: a four-batch layout is a
        # 4-tuple like the (W, S, R, D) of a dynamic variant)
        hit = self._OC_WAVES.get((f64, C), {}).get(
            ('L',) + v.L if v.L else tuple(v)[:4])
        if hit and not ngrad:
            return hit
        w = 2 if f64 else 1
        need = (w * C + 1) * 0 + v.S * (w + 1) + v.R * (5 * w * C + 1) \
            + 8 * w * C + 24 + (6 * w * v.R if ngrad else 0)
        for n in (8, 6, 5, 4, 3, 2):
            if need <= (512 // n) // 8 * 8:
                return n
        return 1

    def _entry_point(self, v, C, nodal=False, tab=False, ngrad=False,
                     maximin=False):
        if v == TABLES:
            return Template(r'''
extern "C" __global__ __launch_bounds__(256)
void ${name}(params_t prm) {
    graphdot::mgk::fill_tables<real_t, ${C}>(prm);
}
''').render(name=self.kernel_name(v, C), C=C)
        if isinstance(v, OCVariant):
            return Template(r'''
extern "C" __global__ __launch_bounds__(${threads})
__attribute__((amdgpu_waves_per_eu(${waves})))
void ${name}(${params} prm) {
    using solver = graphdot::mgk::oc_solver<real_t, ${S}, ${R}, ${W}, ${C},
        ${nodal}, ${D}, ${tab}, ${ngrad}, ${maximin}, ${layout}, graph_t,
        node_kernel_t, edge_kernel_t, p_start_t>;
    static_assert(solver::DLDS == ${dlds}, "Jacobi diagonals in LDS: the host sized the [Y] region for the other answer (HIPBackend.diagonals_in_lds)");
    __shared__ typename solver::lds_t lds;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    solver::run(prm, lds, reinterpret_cast<real_t *>(dyn_lds));
}
''').render(threads=64 * v.W,
            name=self.kernel_name(v, C, nodal, tab, ngrad, maximin),
            dlds='true' if self.diagonals_in_lds(v, C, nodal or ngrad
                                                 or bool(maximin)) else 'false',
            maximin='true' if maximin else 'false',
            layout=('graphdot::mgk::seg_layout<%s>' % ', '.join(map(str, v.L))
                    if v.L else 'graphdot::mgk::dynamic_layout'),
            S=v.S, R=v.R, W=v.W, C=C, D=v.D if v.S else 1,
            waves=self._oc_waves(v, C, ngrad) if ngrad
            else self._waves_without_lds_diagonals(v, C, nodal or bool(maximin)),
            nodal='true' if nodal else 'false',
            tab='true' if tab else 'false',
            ngrad='true' if ngrad else 'false',
            params='params_fd_t' if ngrad else 'params_t')
        if v == GENERAL:
            return Template(r'''
extern "C" __global__ __launch_bounds__(${threads})
void ${name}(params_t prm) {
    using solver = graphdot::mgk::general_solver<real_t, ${threads}, ${C},
        graph_t, node_kernel_t, edge_kernel_t, p_start_t>;
    __shared__ typename solver::lds_t lds;
    solver::run(prm, lds, prm.scratch);
}
''').render(threads=GENERAL_THREADS, name=self.kernel_name(v, C), C=C)
        if v == MFMA:
            return Template(r'''
extern "C" __global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(3)))
void ${name}(params_t prm) {
    using solver = graphdot::mgk::mfma_solver<real_t, graph_t, node_kernel_t,
        edge_kernel_t, p_start_t>;
    __shared__ typename solver::lds_t lds;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    solver::run(prm, lds, dyn_lds);
}
''').render(name=self.kernel_name(v, C))
        if v == STREAM:
            return Template(r'''
extern "C" __global__ __launch_bounds__(${threads})
void ${name}(params_t prm) {
    using solver = graphdot::mgk::stream_solver<real_t, ${threads}, ${C},
        graph_t, node_kernel_t, edge_kernel_t, p_start_t>;
    static_assert(solver::A_ROWS == ${rows} && solver::SEG_CAP == ${cap} && solver::LDS_BUDGET == ${budget}, "rows of p staged per pass, neighbours per segment, LDS budget: host and device disagree");
    __shared__ typename solver::lds_t lds;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    solver::run(prm, lds, dyn_lds, prm.scratch);
}
''').render(threads=STREAM_THREADS, name=self.kernel_name(v, C), C=C,
            rows=STREAM_ROWS[np.dtype(self.real).itemsize], cap=STREAM_CAP,
            budget=STREAM_LDS_BUDGET)
        threads = 64 * v.W * (WPB1 if v.W == 1 else 1)
        return Template(r'''
extern "C" __global__ __launch_bounds__(${threads})
__attribute__((amdgpu_waves_per_eu(${waves})))
void ${name}(params_t prm) {
    using solver = graphdot::mgk::pair_solver<real_t, ${S}, ${R}, ${W}, ${C},
        ${nodal}, ${tab}, graph_t, node_kernel_t, edge_kernel_t, p_start_t>;
    __shared__ typename solver::lds_t lds;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    solver::run(prm, lds, reinterpret_cast<real_t *>(dyn_lds));
}
''').render(threads=threads, name=self.kernel_name(v, C, nodal, tab),
            S=v.S, R=v.R, W=v.W, C=C, waves=self.waves_per_eu(v, C),
            nodal='true' if nodal else 'false',
            tab='true' if tab else 'false')

    def render_source(self, node_kernel, edge_kernel, p, node_t, edge_t,
                      variants, C, nodal=False, tab=False, weighted=False,
                      ngrad=False, maximin=False):
        """Full translation unit for the given solver variants."""
        pd = self._params_dtype(node_kernel, edge_kernel, p)
        pfd = self._params_fd_dtype(node_kernel, edge_kernel, p)
        edge_code = self.gencode_kernel(edge_kernel, 'edge_kernel', self.real)
        edge2_t = ''
        # float builds: the edge microkernel on two records at once where the
        # record and the expression allow it (mgk_oc.h, dense product)
        if (np.dtype(self.real) == np.float32 and packable(edge_t)
                and packed_expression(to_real_expr(
                    edge_kernel.gen_expr('x1', 'x2')[0], 'float32'))):
            edge2_t = '\n' + declstruct2(edge_t, 'edge2_t')
            head = 'struct edge_kernel_t : edge_kernel_theta_t {\n'
            assert head in edge_code
            edge_code = edge_code.replace(
                head, head + '    using packed_edge_t = edge2_t;\n', 1)
        return Template(_TEMPLATE).render(
            real=_real_name(self.real),
            weighted='1' if weighted else '0',
            node_t=declstruct(node_t, 'node_t'),
            edge_t=declstruct(edge_t, 'edge_t') + edge2_t,
            node_kernel=self.gencode_kernel(node_kernel, 'node_kernel',
                                            self.real),
            edge_kernel=edge_code,
            p_start=self.gencode_probability(p, 'p_start', self.real),
            node_size=np.dtype(node_t).itemsize,
            edge_size=max(np.dtype(edge_t).itemsize, 1),
            params_size=pd.itemsize, params_fd_size=pfd.itemsize,
            entry_points=[self._entry_point(v, C, nodal, tab, ngrad, maximin)
                          for v in variants] + [''],
        )

    def _module(self, source):
        # loaded code objects are shared by every backend of the process
        key = jit.cache_key(source, self.hipcc_extra)
        mod = self._modules.get(key)
        if mod is None:
            path = jit.compile_source(source, self.hipcc_extra)
            mod = self._modules[key] = runtime.Module(jit.load_image(path))
        return mod

    # -- job partitioning ---------------------------------------------------------
    @staticmethod
    def _dense_bytes(dgraphs, n):
        """LDS bytes of the two dense edge arrays of a pair (mgk_oc.h DENSE)
        for graphs of at most n nodes -- the largest graph among the pairs of
        the LAUNCH, not of the call: a call that mixes small dense graphs with
        large ones keeps the dense product for the small ones --: n x n
        records, and the record's dword planes in rows of 32 words; 0 if n
        exceeds 32 nodes (the kernel's row stride) or the records are not
        whole words."""
        esize = np.dtype(dgraphs[0].edge_t).itemsize if len(dgraphs) else 0
        if not 0 < n <= 32 or esize == 0 or esize % 4:
            return 0
        return -(-(n * n * esize) // 16) * 16 + esize * (n + 3) * 32 + 256

    @staticmethod
    def stream_virtual_rows(adjacency_count):
        """Lanes the streamed solver deals a graph's nodes onto when it is
        the LDS-resident one (mgk_stream.h, "virtual rows"): a node is cut
        into segments of at most CAP neighbours, CAP = 16 doubled until the
        segments fit the STREAM_THREADS lanes of a workgroup; every node has
        a first segment."""
        d = np.asarray(adjacency_count, dtype=np.int64)
        cap = STREAM_CAP
        while True:
            nv = len(d) + sum(int((d > l * cap).sum())
                              for l in range(1, int(d.max(initial=0)) // cap + 1))
            if nv <= STREAM_THREADS or cap > (1 << 20):
                return nv
            cap *= 2

    def stream_lds_bytes(self, image1, image2, n1, n2, nv1, nv2):
        """Dynamic LDS of a pair in the streamed solver (mgk_stream.h): the
        image of B -- the smaller image among the graphs of at most
        STREAM_THREADS nodes, ties: graph 2 -- in 16-byte units (if it fits
        STREAM_LDS_BUDGET beside the staged rows at their largest; otherwise
        B is read from L2 and takes no LDS), and behind it, for each of the
        G = STREAM_THREADS // ceil64(nV) row groups of a workgroup step (nV:
        B's virtual rows), STREAM_ROWS rows of p and one partial sum per
        lane.  A huge number if neither graph qualifies."""
        image1, image2 = np.asarray(image1), np.asarray(image2)
        n1, n2 = np.asarray(n1), np.asarray(n2)
        w1, w2 = -(-image1 // 16), -(-image2 // 16)
        ok1, ok2 = n1 <= STREAM_THREADS, n2 <= STREAM_THREADS
        first = ok1 & (~ok2 | (w1 < w2))
        nB = np.maximum(np.where(first, n1, n2), 1)
        nV = np.maximum(np.where(first, nv1, nv2), 1)
        LB = (-(-nV // 64) * 64).clip(max=STREAM_THREADS)
        G = STREAM_THREADS // LB
        rs = np.dtype(self.real).itemsize
        stage = (STREAM_ROWS[rs] * G * nB + G * LB) * rs
        image = np.where(first, w1, w2) * 16
        # (an image that does not fit beside the staged rows at their largest
        # stays in L2: the pair's body is instantiated for that case too)
        fits = image + (STREAM_ROWS[rs] + 1) * STREAM_THREADS * rs \
            <= STREAM_LDS_BUDGET
        out = np.where(fits, image, 0) + -(-stage // 16) * 16
        return np.where(ok1 | ok2, out, np.iinfo(np.int64).max // 4)

    def lds_slot_bytes(self, v, C, nodal=False):
        """LDS bytes of the slot values a variant keeps in LDS instead of
        registers (mgk_oc.h SL, `lds_slot_count`: the 16-wave double
        graph-level value solvers, 10 slots per lane; nodal outputs -- and
        with them the nodal-gradient and maximin launches -- keep every slot
        in registers)."""
        if (isinstance(v, OCVariant) and not v.L and C == 1 and v.W == 16
                and not nodal
                and v.S in (40, 64) and np.dtype(self.real) == np.float64):
            return 10 * 64 * v.W * 8
        return 0

    def _waves_without_lds_diagonals(self, v, C, nodal):
        """Occupancy target of a kernel: `waves_per_eu`, but the six-batch
        double value layout runs three waves only WITH its Jacobi diagonals in
        LDS (mgk_oc.h DLDS) -- its nodal flavours and the builds without them
        stay at two (twelve values reloaded per iteration at three)."""
        w = self.waves_per_eu(v, C)
        if (isinstance(v, OCVariant) and v.L == (16, 4, 4, 3, 1, 1) and C == 1
                and np.dtype(self.real) == np.float64
                and not self.diagonals_in_lds(v, C, nodal)
                and not (self.occupancy and (v.W, v.S) in self.occupancy)):
            w = min(w, 2)
        return w

    def diagonals_in_lds(self, v, C, nodal=False):
        """mgk_oc.h DLDS: the double one-wave static value solver of six row
        batches keeps the Jacobi diagonal and its inverse in lane-private LDS
        cells (2 R reals per lane in the [Y] region)."""
        return bool(isinstance(v, OCVariant) and v.L and v.W == 1
                    and v.R == 6 and C == 1 and not nodal
                    and np.dtype(self.real) == np.float64)

    def lds_bytes(self, v, C, ntask=0, gbytes=0, tab_bytes=0, nodal=False):
        """LDS bytes of one workgroup: static p + scratch, dynamic U, the two
        staged graph images per pair slot and the microkernel tables."""
        if isinstance(v, OCVariant):
            # p (rows with the odd stride + the dump cell), the lane-private
            # row sums, the row map, both images; static: tables + scratch
            rs = np.dtype(self.real).itemsize
            NR = 64 * v.W * v.R
            pcap = -(-(np.asarray(ntask) + 1) // 4) * 4
            # static layouts keep the row sums in registers: no Y region,
            # except the value + gradient solvers, which keep x there
            NR_y = 0 if ((v.L and C != 2) or v.S == 0) else NR
            if self.diagonals_in_lds(v, C, nodal):
                NR_y = 2 * NR
            return (pcap + NR_y) * C * rs + 4 * NR + 2 * np.asarray(gbytes) \
                + 4 * v.W * rs + 4 * (128 if v.D > 6 else 64) + 256 + 16 \
                + self.lds_slot_bytes(v, C, nodal)
        wpb = WPB1 if v.W == 1 else 1
        T = 64 * v.W
        ucap = -(-np.asarray(ntask) // 64) * 64 + 64
        return ((v.R * T + ucap) * C * wpb + wpb * 2 * v.W) \
            * np.dtype(self.real).itemsize + wpb * 2 * np.asarray(gbytes) \
            + tab_bytes

    @staticmethod
    def slots_needed(nnz1, n2, jj, deg_sorted, W):
        """Register slots per lane that the stage-1 walk of mgk_solver.h takes
        for each job: tasks (a, i2) are dealt i2-major in batches of T = 64 W,
        and wave w of batch k walks max(1, deg2(i2 of its first task)) slots.
        Returns the maximum over the W waves."""
        T = 64 * W
        ldu = nnz1 + 1                     # padded task stride (mgk_solver.h)
        ntask = ldu * n2
        worst = np.zeros(len(nnz1), dtype=np.int64)
        nb = int(-(-ntask.max() // T)) if len(ntask) else 0
        for w in range(W):
            total = np.zeros(len(nnz1), dtype=np.int64)
            for k in range(nb):
                first = k * T + 64 * w
                live = first < ntask
                if not live.any():
                    break
                i2 = np.where(live, first // ldu, 0)
                d = np.maximum(deg_sorted[jj, i2], 1)
                total += np.where(live, d, 0)
            worst = np.maximum(worst, total)
        return worst

    @staticmethod
    def oc_trips(hist1, hist2, D, nb):
        """Per-batch trip counts of the one-wave owner-computes walk: entry
        [t, k] is the degree product of the first row of batch k (rows
        64 k ...) of job t in the sorted row order, 0 for batches without
        rows.  A static layout L fits job t iff trips[t, k] <= L[k] for all
        k < len(L) and the job has no rows beyond 64 len(L)."""
        order = sorted(((a, b) for a in range(D + 1) for b in range(D + 1)),
                       key=lambda t: -t[0] * t[1])
        prods = np.array([a * b for a, b in order] + [0], dtype=np.int64)
        sizes = hist1[:, [a for a, _ in order]] * hist2[:, [b for _, b in order]]
        cum = np.cumsum(sizes, axis=1)                   # (n, ncp)
        n = len(cum)
        if n == 0:
            return np.zeros((0, nb), dtype=np.int64)
        first = 64 * np.arange(nb, dtype=np.int64)
        # rectangle of row `first`: number of cumulative sizes <= first
        c = (cum[:, None, :] <= first[None, :, None]).sum(axis=2)
        live = first[None, :] < cum[:, -1:]
        return np.where(live, prods[np.minimum(c, len(order))], 0)

    @staticmethod
    def oc_slots_needed(hist1, hist2, W, D):
        """Slots per lane of the owner-computes walk (mgk_oc.h): rows are
        sorted by descending degree product -- rectangle (d1, d2) holds
        hist1[d1] * hist2[d2] rows -- and dealt in batches of T = 64 W, the
        64-row chunks of a batch to the waves in snake order; a wave walks, per
        batch, the product of the first row of its chunk.  `hist*`: (n_jobs,
        D + 1) degree histograms of the two graphs.  Returns the maximum over
        the waves."""
        order = sorted(((a, b) for a in range(D + 1) for b in range(D + 1)),
                       key=lambda t: -t[0] * t[1])    # stable: row-major ties
        ncp = len(order)
        prods = np.array([a * b for a, b in order] + [0], dtype=np.int64)
        sizes = hist1[:, [a for a, _ in order]] * hist2[:, [b for _, b in order]]
        cum = np.cumsum(sizes, axis=1)
        n = len(cum)
        if n == 0:
            return np.zeros(0, dtype=np.int64)
        N = cum[:, -1]
        T = 64 * W
        nb = int(-(-N.max() // T))
        # rectangle of the first row of every (wave, batch): one sorted search
        # over all jobs at once (row r of `cum` shifted by r * stride)
        stride = int(max(N.max(), nb * T)) + 1
        shift = np.arange(n, dtype=np.int64) * stride
        flat = (cum + shift[:, None]).ravel()
        worst = np.zeros(n, dtype=np.int64)
        kk = np.arange(nb, dtype=np.int64)
        for w in range(W):
            # (snake order of the chunks over the waves: mgk_oc.h, row_pos)
            first = kk * T + 64 * np.where(kk % 2 == 1, W - 1 - w, w)
            c = np.searchsorted(flat, (first[None, :] + shift[:, None]).ravel(),
                                side='right').reshape(n, nb) \
                - (np.arange(n, dtype=np.int64) * ncp)[:, None]
            live = first[None, :] < N[:, None]
            trip = np.where(live, prods[np.minimum(c, ncp)], 0)
            worst = np.maximum(worst, trip.sum(axis=1))
        return worst

    def classify(self, ji, jj, dgraphs, C, tab_bytes=0, gtab=False,
                 oc_only=False, nodal=False, mfma=False):
        """Assign every job the cheapest solver variant it fits.  Returns
        (variant index, cost, stage-1 tasks, image bytes, padded rows, image
        bytes incl. class ids) per job.

        Everything the assignment looks at -- node and nonzero counts, image
        sizes, the slot walks -- is a function of the two graphs' degree
        histograms (nodes are stored by descending degree) and image sizes,
        and a set of graphs has few distinct ones (about 150 among the 1000
        QM7-like molecules): large job lists are classified once per pair of
        graph classes and looked up."""
        sel, out = self._classify_classes(ji, jj, dgraphs, C, tab_bytes, gtab,
                                          oc_only, nodal=nodal, mfma=mfma)
        out = out[:6]      # (the seventh, the larger node count, is _partition's)
        return out if sel is None else tuple(a[sel.sel] for a in out)

    def _classify_classes(self, ji, jj, dgraphs, C, tab_bytes=0, gtab=False,
                          oc_only=False, jobs=None, nodal=False, mfma=False):
        """(sel, per-class-pair results): job t has the results of class pair
        sel.sel[t] (`ClassPairs`: class-pair key per job, jobs per key); sel is
        None for short job lists (results are per job).  `jobs`: the job list
        as the (u32, u32) record array ji, jj were taken from (native path)."""
        n_jobs = len(jobs) if jobs is not None else len(ji)
        if n_jobs < 4096:
            if ji is None:
                ji, jj = jobs['i'], jobs['j']
            ji = np.asarray(ji, dtype=np.int64)
            jj = np.asarray(jj, dtype=np.int64)
            return None, self._classify_pairs(ji, jj, dgraphs, C, tab_bytes,
                                              gtab, oc_only, nodal, mfma)
        # graphs of one degree histogram and image size are classified alike
        f = graph_features(dgraphs)
        if int(f['max_degree'].max()) < HIST_BINS - 1:
            # (the 16-bin histograms of the headers are exact)
            key = np.column_stack((f['hist'], f['image_bytes']))
        else:
            width = int(f['max_degree'].max()) + 1
            key = np.zeros((len(dgraphs), width + 1), dtype=np.int64)
            for k, g in enumerate(dgraphs):
                key[k, :width] = np.bincount(g.adjacency_count,
                                             minlength=width)
                key[k, width] = g.image_bytes
        if self.native:
            # (distinct rows through a hash table, gdh_number_records:
            # np.unique(axis=0) sorts the rows as byte strings, 0.7 ms)
            from ...hip import hostlib
            rows = np.ascontiguousarray(key)
            rows = rows.view(np.dtype((np.void, rows.shape[1]
                                       * rows.itemsize))).reshape(-1)
            cid, rep = hostlib.number_records(rows, [(0, rows.dtype.itemsize)])
        else:
            _, rep, cid = np.unique(key, axis=0, return_index=True,
                                    return_inverse=True)
        cid, nc = cid.reshape(-1).astype(np.int32), len(rep)
        if self.native and jobs is not None:
            from ...hip import hostlib
            pk, count = hostlib.pair_keys(jobs, cid, nc)
        else:
            if ji is None:
                ji, jj = jobs['i'], jobs['j']
            ji = np.asarray(ji, dtype=np.int64)
            jj = np.asarray(jj, dtype=np.int64)
            pk = cid[ji] * np.int32(nc) + cid[jj]
            count = np.bincount(pk, minlength=nc * nc)
        upk = np.flatnonzero(count)
        out = self._classify_pairs(rep[upk // nc], rep[upk % nc], dgraphs, C,
                                   tab_bytes, gtab, oc_only, nodal, mfma)
        return ClassPairs(pk, upk, count, nc), out

    #: row batches the trip tables cover (static layouts have at most this many)
    TRIP_BATCHES = 12

    def _degree_hists(self, dgraphs, maxdeg, D, hists):
        """(distinct degree histograms H, id of every graph's) for the graphs
        of largest degree <= D (others: a zero row)."""
        if D not in hists:
            h = np.zeros((len(dgraphs), D + 1), dtype=np.int64)
            for g_, dg_ in enumerate(dgraphs):
                if maxdeg[g_] <= D:
                    h[g_] = np.bincount(dg_.adjacency_count, minlength=D + 1)
            H, hid = np.unique(h, axis=0, return_inverse=True)
            hists[D] = (H, hid.reshape(-1))
        return hists[D]

    def _trip_table(self, ji, jj, dgraphs, maxdeg, pair_maxdeg, D, hists):
        """(trips per distinct histogram pair in use, row of every job in
        that table or -1 if a graph of the job exceeds degree D)."""
        H, hid = self._degree_hists(dgraphs, maxdeg, D, hists)
        nH = len(H)
        row = np.full(len(ji), -1, dtype=np.int64)
        idx = np.flatnonzero(pair_maxdeg <= D)
        pk = hid[ji[idx]] * nH + hid[jj[idx]]
        upk, inv = np.unique(pk, return_inverse=True)
        row[idx] = inv.reshape(-1)
        tr = self.oc_trips(H[upk // nH], H[upk % nH], D, self.TRIP_BATCHES)
        return tr, row

    def _classify_pairs(self, ji, jj, dgraphs, C, tab_bytes=0, gtab=False,
                        oc_only=False, nodal=False, mfma=False):
        f = graph_features(dgraphs)
        n_node, n_nz = f['n_node'], f['n_nz']
        deg_sorted = None      # (the two-stage variants' walk: made on demand)
        n1, n2 = n_node[ji], n_node[jj]
        nnz1 = n_nz[ji]
        N = n1 * n2
        NP = n1 * (n2 | 1)                 # row space with the odd LDS stride
        cost = n_nz[ji] * n_nz[jj] + 4 * N
        choice = np.full(len(ji), -1, dtype=np.int64)
        slots = {}
        # U entries per pair: one per stage-1 task; the region also stages the
        # CSR row pointers of both graphs during setup
        ntask = np.maximum((nnz1 + 1) * n2, n1 + n2 + 2)
        image = f['image_bytes']
        if tab_bytes:      # the label-class section is staged with the image
            image = image + class_bytes(n_node, n_nz)
        gbytes = np.maximum(image[ji], image[jj])
        # the owner-computes solvers with global tables stage the class ids
        image_oc = image + class_bytes(n_node, n_nz) if gtab else image
        gbytes_oc = np.maximum(image_oc[ji], image_oc[jj])
        maxdeg = f['max_degree']
        pair_maxdeg = np.maximum(maxdeg[ji], maxdeg[jj])
        oc_slots, hists, trips = {}, {}, {}
        # (the on-the-fly kernels have every flavour of the slot kernels but
        # the static layouts)
        fly_off = False
        # `rem`: the jobs without a variant yet -- every test below runs on
        # that shrinking subset only (most jobs leave in the first variants)
        rem = np.arange(len(ji))
        # the owner-computes menu natively (gdh_classify_oc restates the
        # tests of the loop below), when it precedes every other variant
        is_oc = [isinstance(v, OCVariant) for v in self.variants]
        n_oc = sum(is_oc)
        native_oc = (self.native and not tab_bytes and n_oc > 0
                     and all(is_oc[:n_oc]) and len(ji) > 0
                     and len(dgraphs) < 2**31)
        if native_oc:
            from ...hip import hostlib
            menu = [(k, v) for k, v in enumerate(self.variants[:n_oc])
                    if (not v.L or self._static_enabled(v, C))
                    and (v.S > 0 or not fly_off)]
            hist = f['hist']
            ch, _ = hostlib.classify_oc(
                ji, jj, n_node, n_nz, image_oc, maxdeg, hist,
                [(v.W, v.S, v.R, v.D, v.L) for _, v in menu], C,
                np.dtype(self.real).itemsize, LDS_LIMIT, FLY_MIN_DEGREE,
                [self.lds_slot_bytes(v, C, nodal) for _, v in menu])
            idx = np.array([k for k, _ in menu], dtype=np.int64)
            hit = ch >= 0
            choice[hit] = idx[ch[hit]]
            rem = rem[~hit]
        for k, v in enumerate(self.variants):
            if not len(rem):
                break
            if v in (GENERAL, STREAM, MFMA) or (
                    oc_only and not isinstance(v, OCVariant)):
                continue
            if isinstance(v, OCVariant):
                if tab_bytes or native_oc:  # (table kernels: two-stage only)
                    continue
                if v.L and not self._static_enabled(v, C):
                    continue
                if v.S == 0:
                    # on-the-fly: the pairs whose degrees no slot variant
                    # takes (sparse pairs that merely overflow
                    # the slots are faster in the two-stage solver: 152 against
                    # 229 us for the 118 largest pairs of configuration 2)
                    if fly_off:
                        continue
                    fits = (N[rem] <= 64 * v.W * v.R) & (NP[rem] < 0x3FFF) \
                        & (pair_maxdeg[rem] > FLY_MIN_DEGREE)
                    fits &= self.lds_bytes(v, C, NP[rem], gbytes_oc[rem],
                                           nodal=nodal) <= LDS_LIMIT
                    choice[rem[fits]] = k
                    rem = rem[~fits]
                    continue
                fits = ((pair_maxdeg[rem] <= v.D) & (N[rem] <= 64 * v.W * v.R)
                        & (NP[rem] < 0xFFFF))
                if not fits.any():
                    continue
                fits &= self.lds_bytes(v, C, NP[rem], gbytes_oc[rem],
                                       nodal=nodal) <= LDS_LIMIT
                if v.L:
                    # static layout: the trip count of every batch under its
                    # segment (looked up per pair of distinct histograms)
                    if v.D not in trips:
                        trips[v.D] = self._trip_table(
                            ji, jj, dgraphs, maxdeg, pair_maxdeg, v.D, hists)
                    tr, row = trips[v.D]
                    idx = rem[fits]
                    ok = row[idx] >= 0
                    L = np.zeros(tr.shape[1], dtype=np.int64)
                    L[:v.R] = v.L
                    ok[ok] = (tr[row[idx[ok]]] <= L[None, :]).all(axis=1)
                    choice[idx[ok]] = k
                    fits[fits] = ok
                    rem = rem[~fits]
                    continue
                if (v.W, v.D) not in oc_slots:
                    # the walk depends on the two degree histograms only,
                    # and a set of graphs has few distinct ones: it is
                    # evaluated once per pair of distinct histograms
                    H, hid = self._degree_hists(dgraphs, maxdeg, v.D, hists)
                    idx = rem[pair_maxdeg[rem] <= v.D]
                    pk = hid[ji[idx]] * len(H) + hid[jj[idx]]
                    seen = np.zeros(len(H) * len(H), dtype=bool)
                    seen[pk] = True
                    upk = np.flatnonzero(seen)
                    lut = np.zeros(len(H) * len(H), dtype=np.int64)
                    lut[upk] = self.oc_slots_needed(
                        H[upk // len(H)], H[upk % len(H)], v.W, v.D)
                    sl = np.full(len(ji), np.iinfo(np.int64).max,
                                 dtype=np.int64)
                    sl[idx] = lut[pk]
                    oc_slots[(v.W, v.D)] = sl
                fits &= oc_slots[(v.W, v.D)][rem] <= v.S
                choice[rem[fits]] = k
                rem = rem[~fits]
                continue
            fits = (NP[rem] <= 64 * v.W * v.R) & (NP[rem] <= 0xFFFF)
            if not fits.any():
                continue
            fits &= self.lds_bytes(v, C, ntask[rem], gbytes[rem], tab_bytes) \
                <= LDS_LIMIT
            if v.W not in slots:
                if deg_sorted is None:
                    deg_sorted = np.zeros((len(dgraphs), int(n_node.max())),
                                          dtype=np.int64)
                    for k_, g in enumerate(dgraphs):
                        deg_sorted[k_, :g.n_node] = g.adjacency_count
                sl = np.full(len(ji), np.iinfo(np.int64).max, dtype=np.int64)
                sl[rem] = self.slots_needed(nnz1[rem], n2[rem], jj[rem],
                                            deg_sorted, v.W)
                slots[v.W] = sl
            fits &= slots[v.W][rem] <= v.S
            choice[rem[fits]] = k
            rem = rem[~fits]
        if mfma and MFMA in self.variants and C == 1 and not oc_only:
            # dense pairs of small graphs under a label-blind edge kernel: the
            # dense-tile solver on the matrix cores (mgk_mfma.h) -- by the
            # rule of the dense product of the on-the-fly variants (mgk_oc.h
            # DENSE): adjacency matrices more than ~60 % full
            dense = (n1 <= MFMA_MAX_NODES) & (n2 <= MFMA_MAX_NODES) & \
                (23 * n_nz[ji] * n_nz[jj] > 9 * N * N)
            choice[dense] = self.variants.index(MFMA)
            gbytes = np.where(dense, np.maximum(f['image_bytes'][ji],
                                                f['image_bytes'][jj]), gbytes)
        if np.any(choice < 0) and oc_only:
            raise NotOwnerComputes
        if np.any(choice < 0):
            if GENERAL not in self.variants:
                bad = int(np.argmax(choice < 0))
                raise NotImplementedError(
                    f'graph pair ({ji[bad]}, {jj[bad]}) with '
                    f'{n1[bad]}x{n2[bad]} nodes and {nnz1[bad]}x'
                    f'{n_nz[jj[bad]]} adjacency nonzeros exceeds the largest '
                    'register-resident solver variant and the general '
                    'solver is disabled')
            left = choice < 0
            choice[left] = self.variants.index(GENERAL)
            if STREAM in self.variants:
                # large pairs (values, and values + analytic gradient as two
                # sequential solves): the streamed solver, if the
                # image of the smaller graph (of at most STREAM_THREADS nodes:
                # one lane per node of it) fits the LDS beside the staged
                # rows of p -- the rule of mgk_stream.h
                raw = f['image_bytes']
                # (virtual rows of the graphs these pairs touch: a pass over
                # their degree lists, only when there are such pairs)
                nv = np.zeros(len(dgraphs), dtype=np.int64)
                for g_ in np.unique(np.concatenate((ji[left], jj[left]))):
                    nv[g_] = self.stream_virtual_rows(
                        dgraphs[g_].adjacency_count) \
                        if n_node[g_] <= STREAM_THREADS else 0
                sb = self.stream_lds_bytes(raw[ji], raw[jj], n1, n2,
                                           nv[ji], nv[jj])
                fits = left & (sb + 1024 <= LDS_LIMIT)
                choice[fits] = self.variants.index(STREAM)
                # (for these pairs `gbytes` is the dynamic LDS of the pair)
                gbytes = np.where(fits, sb, gbytes)
        return choice, cost, ntask, gbytes, NP, gbytes_oc, np.maximum(n1, n2)

    # -- the three phases -----------------------------------------------------------
    def _graphs_and_kernels(self, graphs, node_kernel, edge_kernel, traits,
                            timer=None, ngrad=False):
        """Pack (or fetch the cached packing of) every graph; wrap the edge
        kernel for weighted graphs; pick the solver flavour C."""
        tic = timer.tic if timer else (lambda *_: None)
        toc = timer.toc if timer else (lambda *_: None)
        tic('transferring graphs to GPU')
        # graphs seen for the first time are packed together in one
        # vectorised pass (_devicegraph.pack_many)
        # (the same list of graphs as in a recent call, none touched since:
        # the same packings -- util.cookie.IdentityCache)
        if not isinstance(graphs, (list, tuple)):
            graphs = list(graphs)
        ckey, dgraphs = self._dgraph_lists.get(graphs)
        if dgraphs is None:
            key = (self.uuid.int, np.dtype(self.real).str)   # (an int hashes in C)
            new = [g for g in graphs if key not in g.cookie]
            if new:
                for g, dg in zip(new, pack_many(new, real=self.real,
                                                native=self.native)):
                    g.cookie[key] = dg
            dgraphs = _DeviceGraphList(g.cookie[key] for g in graphs)
            sig0 = dgraphs[0].signature
            for dg in dgraphs:
                if dg.signature != sig0:
                    self._assert_homogeneous(dgraphs[0], dg)
            dgraphs.ids = tuple(map(id, dgraphs))
            self._dgraph_lists.put(ckey, graphs, dgraphs)
        toc('transferring graphs to GPU')
        if traits.eval_gradient is True and traits.nodal is not False \
                and not ngrad:
            raise NotImplementedError(
                'nodal gradients are evaluated by finite differences over '
                'value launches (HIPBackend._nodal_gradient), not by a '
                'gradient plan')
        C = 2 if traits.eval_gradient is True and not ngrad else 1
        # attributes the microkernels read: label classes are numbered over
        # these (before the weighted wrapper: the weight is not a label)
        fields = (self._used_fields(node_kernel),
                  self._used_fields(edge_kernel))
        if dgraphs[0].weighted:
            edge_kernel = TensorProduct(weight=Product(), label=edge_kernel)
        return dgraphs, edge_kernel, C, fields

    def _partition(self, dgraphs, jobs, C, tab_bytes=0, gtab=False,
                   oc_only=False, merge_map=None, nodal=False, mfma=False):
        """Host half of a layout: solver variant per job, launch order (by
        variant, then descending cost) and launch geometry.  No device."""
        jobs = np.ascontiguousarray(jobs)
        jobs_sorted = None             # (set by the native ordering)
        sel, (choice, cost, ntask, gbytes, NP, gbytes_oc, nmax) = \
            self._classify_classes(None, None, dgraphs, C, tab_bytes, gtab,
                                   oc_only, jobs=jobs, nodal=nodal, mfma=mfma)
        # Launch order: by variant, then descending cost, then job index.
        # `choice` ... `gbytes_oc` are per class pair (or per job when sel is
        # None); the jobs are ordered by the rank of their class pair with one
        # stable sort of small integers, and every per-launch maximum is taken
        # over class pairs.
        # Launches of a few thousand waves cannot fill 256 CUs for long (a pair
        # takes ~20 us from staging to result: every launch has a tail of that
        # length): their jobs ride in the next larger owner-computes variant in
        # use (same waves per pair and degree bound, S and R at least as large)
        # -- fewer, fuller launches, which matters most for the shards of a
        # multi-GPU run (scripts/minlaunch_experiment.sh: 8192 against 2048
        # waves: -1 % on the full matrix, -2...8 % on 1/2...1/8 of it).
        applied = {}                   # launch merging of this job list
        if merge_map is not None:
            # decided elsewhere, on a larger job list this one is a shard of
            # (_sharded.ShardPlan.merge_map)
            for k, k2 in merge_map.items():
                choice = np.where(choice == k, k2, choice)
            applied = dict(merge_map)
        elif self.min_launch > 0:
            members_ = np.ones(len(choice), dtype=np.int64) if sel is None \
                else sel.members
            used_ = sorted(set(choice.tolist()))
            for a_, k in enumerate(used_):
                v = self.variants[k]
                if not isinstance(v, OCVariant):
                    continue
                here = choice == k
                if int(members_[here].sum()) * v.W >= self.min_launch:
                    continue
                for k2 in used_[a_ + 1:]:
                    v2 = self.variants[k2]
                    if not (isinstance(v2, OCVariant) and v2.W == v.W
                            and v2.D == v.D and v2.S >= v.S and v2.R >= v.R):
                        continue
                    # a static layout takes the jobs of another static layout
                    # it dominates batch by batch (never those of a dynamic
                    # variant); a dynamic variant takes any total <= S
                    if v2.L and not (v.L and all(
                            a <= b for a, b in zip(v.L, v2.L))):
                        continue
                    if True:
                        choice = np.where(here, k2, choice)
                        # (chains collapse: what rode in k now rides in k2)
                        for k0, k1 in list(applied.items()):
                            if k1 == k:
                                applied[k0] = k2
                        applied[k] = k2
                        break
        choice = self._fit_launches_into_lds(choice, C, NP, ntask, gbytes,
                                             gbytes_oc, tab_bytes, oc_only,
                                             nodal)
        rank_of = np.empty(len(choice), dtype=np.int64)
        by_rank = np.lexsort((-cost, choice))
        # class pairs of equal (variant, cost) share a rank: their jobs stay
        # in job order, as if every job had been sorted by itself
        step = np.ones(len(choice), dtype=np.int64)
        step[1:] = (choice[by_rank[1:]] != choice[by_rank[:-1]]) | \
            (cost[by_rank[1:]] != cost[by_rank[:-1]])
        rank_of[by_rank] = np.cumsum(step) - 1
        if sel is None:
            order_all = by_rank.astype(np.uint32)   # (lexsort is stable)
            members = np.ones(len(choice), dtype=np.int64)
        elif self.native:
            # stable counting sort of the jobs by the rank of their class pair
            from ...hip import hostlib
            rank_of_key = np.full(sel.nc * sel.nc, -1, dtype=np.int32)
            rank_of_key[sel.upk] = rank_of
            order_all, jobs_sorted = hostlib.order_jobs(
                sel.pk, rank_of_key, int(rank_of.max()) + 1, jobs)
            members = sel.members
        else:
            rank = rank_of[sel.sel]
            rank = rank.astype(np.uint16 if len(choice) <= 0xFFFF
                               else np.uint32)
            order_all = np.argsort(rank, kind='stable').astype(np.uint32)
            members = sel.members
        used = sorted(set(choice.tolist()))
        rsize = np.dtype(self.real).itemsize
        launches, cursor = [], 0
        for k in used:
            v = self.variants[k]
            idx = np.flatnonzero(choice == k)      # class pairs (or jobs)
            count = int(members[idx].sum())
            if v == GENERAL:
                # one workgroup per pair, CG vectors + U in global scratch
                # (rows N <= padded rows NP)
                per_wg = int(((3 * NP[idx] + ntask[idx]) * C).max())
                launches.append(dict(
                    variant=v, k=k, offset=cursor, ucap=per_wg, gcap=0,
                    dynamic_lds=0, count=count, grid=None,
                    threads=GENERAL_THREADS, per_wg=per_wg))
                cursor += count
                continue
            if v == MFMA:
                # one wave per pair; dynamic LDS: the two images
                gcap = int(-(-gbytes[idx].max() // 16) * 16)
                launches.append(dict(
                    variant=v, k=k, offset=cursor, ucap=0, gcap=gcap,
                    dynamic_lds=2 * gcap, count=count, grid=count,
                    threads=64))
                cursor += count
                continue
            if v == STREAM:
                # one to 256 workgroups per pair; scratch [x | r | p | Ap |
                # diag (| second solution)] (N <= NP reals each); LDS: image
                # of B | staged rows of p | partial sums
                per_wg = int((6 if C == 2 else 5) * NP[idx].max())
                dyn = int(-(-gbytes[idx].max() // 16) * 16)
                # (the pairs' costs in launch order, largest first: `prepare`
                # picks the workgroups per pair from them)
                # (a tuple: launch descriptions are compared as values)
                costs = tuple(np.sort(np.repeat(cost[idx], members[idx])
                                      .astype(np.float64))[::-1].tolist())
                launches.append(dict(
                    variant=v, k=k, offset=cursor, ucap=per_wg, gcap=dyn,
                    dynamic_lds=dyn, count=count, costs=costs,
                    grid=None, threads=STREAM_THREADS, per_wg=per_wg))
                cursor += count
                continue
            if isinstance(v, OCVariant):
                # one pair per workgroup; dynamic LDS: p | row sums | row map
                # | two images (mgk_oc.h)
                pcap = int(-(-(NP[idx].max() + 1) // 4) * 4)
                gcap = int(-(-gbytes_oc[idx].max() // 16) * 16)
                NR = 64 * v.W * v.R
                NR_y = 0 if ((v.L and C != 2) or v.S == 0) else NR
                if self.diagonals_in_lds(v, C, nodal):
                    NR_y = 2 * NR
                dyn = (pcap + NR_y) * C * rsize + 4 * NR + 2 * gcap
                dyn += self.lds_slot_bytes(v, C, nodal)
                dense = False
                if v.S == 0:
                    # on-the-fly variants: the dense n x n edge-record arrays
                    # of both graphs (mgk_oc.h DENSE), sized for the largest
                    # graph of the call -- when they fit (F_DENSE tells the
                    # kernel that they are there)
                    extra = self._dense_bytes(dgraphs, int(nmax[idx].max()))
                    if extra and dyn + extra + 4096 <= LDS_LIMIT:
                        # (+ 4 cells of p: the last trip of a row of the
                        # dense product reads up to three cells past it)
                        pcap += 4
                        dyn += extra + 4 * C * rsize
                        dense = True
                launches.append(dict(
                    variant=v, k=k, offset=cursor, ucap=pcap, gcap=gcap,
                    dynamic_lds=dyn, count=count,
                    grid=int(max(1, -(-count // self.jobs_per_unit))),
                    threads=64 * v.W, tab=gtab, dense=dense))
                cursor += count
                continue
            wpb = WPB1 if v.W == 1 else 1
            threads = 64 * v.W * wpb
            # Many small workgroups (jobs_per_unit pairs per wave, default
            # one) rather than one persistent grid: the hardware dispatcher
            # then balances the load whatever the achieved residency is, and
            # a launch of few jobs (a rare variant, one rank's shard of a
            # multi-GPU run) still covers the chip.  Measured: 1 pair per wave
            # = 8 per wave + 1 % on 500 500 pairs, + 28 % on 61 000.
            per_unit = self.jobs_per_unit
            grid = int(max(1, -(-count // (wpb * per_unit))))
            ucap = int(-(-ntask[idx].max() // 64) * 64) + 64   # + zero pad
            gcap = int(-(-gbytes[idx].max() // 16) * 16)
            dyn = (ucap * C * rsize + 2 * gcap) * wpb + tab_bytes
            launches.append(dict(variant=v, k=k, offset=cursor, ucap=ucap,
                                 gcap=gcap, dynamic_lds=dyn, count=count,
                                 grid=grid, threads=threads))
            cursor += count
        out = Partition((jobs, used, order_all, launches))
        out.jobs_sorted, out.merge_map = jobs_sorted, applied
        return out

    def _launch_lds(self, v, C, NP, ntask, gbytes, gbytes_oc, tab_bytes,
                    nodal=False):
        """LDS of a workgroup -- static and dynamic, `lds_bytes` -- of ONE
        launch of variant `v` over the given class pairs (or jobs): the
        regions are sized for the largest vector and the largest image among
        them -- two maxima that need not belong to one pair."""
        if isinstance(v, OCVariant):
            return int(self.lds_bytes(v, C, int(NP.max()),
                                      int(gbytes_oc.max()), nodal=nodal))
        return int(self.lds_bytes(v, C, int(ntask.max()), int(gbytes.max()),
                                  tab_bytes))

    def _fit_launches_into_lds(self, choice, C, NP, ntask, gbytes, gbytes_oc,
                               tab_bytes, oc_only, nodal=False):
        """Every pair was given a variant whose LDS regions hold IT; a launch
        sizes its regions for the largest vector and the largest graph image
        among its pairs, and the two can come from different pairs: the sum
        went 0.5 KB over the 160 KB of a CU for the 16-wave double variant
        with LDS-resident slot values on 45...62-node graphs, and 2.4 KB over
        for the 16-wave two-stage variant (64 KB of it static) on a list of
        170 mixed graphs -- invalid launches, scripts/fuzz_parity.py seeds 23
        and 51.  Pairs that hold one of the two maxima of such a launch leave
        for the general solver until the launch fits."""
        limit = LDS_LIMIT
        fallback = None
        for k in sorted(set(choice.tolist())):
            v = self.variants[k]
            if k < 0 or v in (GENERAL, STREAM, MFMA):
                continue
            while True:
                idx = np.flatnonzero(choice == k)
                if not len(idx) or self._launch_lds(
                        v, C, NP[idx], ntask[idx], gbytes[idx],
                        gbytes_oc[idx], tab_bytes, nodal) <= limit:
                    break
                if fallback is None:
                    if oc_only or GENERAL not in self.variants:
                        raise NotOwnerComputes(
                            f'the pairs of variant {v} do not fit the LDS '
                            'of a compute unit together and the general '
                            'solver is not available to this call')
                    fallback = self.variants.index(GENERAL)
                big_p = NP[idx] if isinstance(v, OCVariant) else ntask[idx]
                big_g = gbytes_oc[idx] if isinstance(v, OCVariant) \
                    else gbytes[idx]
                # drop whichever maximum frees more bytes per pair dropped
                cand = []
                for key in (big_p, big_g):
                    out = idx[key == key.max()]
                    keep = np.setdiff1d(idx, out)
                    after = self._launch_lds(
                        v, C, NP[keep], ntask[keep], gbytes[keep],
                        gbytes_oc[keep], tab_bytes, nodal) if len(keep) else 0
                    cand.append((after, len(out), out))
                after, _, out = min(cand, key=lambda c: (c[0], c[1]))
                choice = choice.copy()
                choice[out] = fallback
        return choice

    @staticmethod
    def _code_signature(node_kernel, edge_kernel, p, dgraphs, C, nodal,
                        tab=False, gtab=False, ngrad=False, maximin=False):
        """What the generated code depends on: the microkernel expressions
        and record types (not the hyperparameter values), the graphs' record
        types and the output mode.  (The dtypes themselves, not their
        strings: numpy takes 15 us to print a structured dtype.)"""
        return (node_kernel.gen_expr('x1', 'x2')[0], np.dtype(node_kernel.dtype),
                edge_kernel.gen_expr('x1', 'x2')[0], np.dtype(edge_kernel.dtype),
                p.gen_expr()[0], np.dtype(p.dtype), dgraphs[0].signature,
                C, nodal, tab, gtab, ngrad, maximin)

    def _sources(self, used, node_kernel, edge_kernel, p, dgraphs, C, nodal,
                 tab=False, gtab=False, ngrad=False, maximin=False):
        """One translation unit per solver variant in use (+ the one of the
        table kernel, key 'tables', when the owner-computes solvers read
        global tables).  The node / edge / start-probability code is shared
        text; only the entry point differs."""
        # rendering is pure text work on hyperparameter-independent inputs:
        # memoised on the generated expressions and the record types
        sig = self._code_signature(node_kernel, edge_kernel, p, dgraphs, C,
                                   nodal, tab, gtab, ngrad, maximin)
        out = {}
        todo = [(k, self.variants[k]) for k in used]
        if gtab and any(isinstance(v, OCVariant) for _, v in todo):
            todo.append(('tables', TABLES))
        opts = (np.dtype(self.real).str, tuple(self.hipcc_extra), WPB1,
                self.tables)
        for k, v in todo:
            # (the entry point also carries the variant's occupancy target)
            waves = None if v in (GENERAL, TABLES, STREAM, MFMA) else (
                self._oc_waves(v, C, ngrad) if ngrad and isinstance(
                    v, OCVariant) else self.waves_per_eu(v, C))
            key = (sig, tuple(v), waves, opts)
            if key not in self._source_cache:
                self._source_cache[key] = self.render_source(
                    node_kernel, edge_kernel, p, dgraphs[0].node_t,
                    dgraphs[0].edge_t, [v], C, nodal,
                    tab=gtab if isinstance(v, OCVariant)
                    else (tab and v not in (GENERAL, TABLES, STREAM, MFMA)),
                    weighted=dgraphs[0].weighted, ngrad=ngrad,
                    maximin=maximin and isinstance(v, OCVariant))
            out[k] = self._source_cache[key]
        return out

    def _label_blind(self, edge_kernel):
        """Is the edge microkernel (as the caller gave it, without the weight
        wrapper) a constant -- separable in one term, what the dense-tile
        MFMA solver takes (mgk_mfma.h)?  Float builds only."""
        return (np.dtype(self.real) == np.float32
                and os.environ.get('GD_MFMA', '1') != '0'
                and getattr(edge_kernel, 'name', None) == 'Constant')

    def _frontend(self, graphs, node_kernel, edge_kernel, p, jobs, traits,
                  timer=None):
        """Host-only half of `prepare` (used by `precompile`): pack graphs,
        partition the jobs and render one translation unit per solver variant
        in use."""
        edge_kernel_in = edge_kernel
        dgraphs, edge_kernel, C, fields = self._graphs_and_kernels(
            graphs, node_kernel, edge_kernel, traits, timer)
        arena = self._host_arena(dgraphs, fields)
        tab_bytes = self._table_bytes(arena)
        gtab = self._global_tables(arena)
        jobs, used, order_all, launches = self._partition(
            dgraphs, jobs, C, tab_bytes, gtab,
            nodal=traits.nodal is not False,
            mfma=self._label_blind(edge_kernel_in))
        sources = self._sources(used, node_kernel, edge_kernel, p, dgraphs, C,
                                traits.nodal is not False, tab_bytes > 0,
                                gtab)
        return dgraphs, edge_kernel, jobs, C, used, order_all, launches, \
            sources

    def precompile(self, graphs, node_kernel, edge_kernel, p, jobs, traits):
        """Compile (into the on-disk JIT cache) every code object that
        `prepare` would need for this call.  Needs hipcc but no device."""
        *_, sources = self._frontend(graphs, node_kernel, edge_kernel, p,
                                     jobs, traits)
        return jit.compile_many(list(sources.values()), self.hipcc_extra)

    def _layout(self, dgraphs, jobs, starts, C, fields=(None, None),
                timer=None, ngrad=False, maximin=False, merge_map=None,
                nodal=False, mfma=False):
        """Everything of a plan that depends only on WHICH pairs of WHICH
        graphs are evaluated: variant per job, launch order and geometry, and
        the device copies of the job list, the order and `starts`.  Cached
        (LRU) on the identity of the packed graphs and a checksum of the job
        list, so that repeated evaluations with new hyperparameters -- the
        training loop of a Gaussian process -- skip the host-side work and
        the uploads."""
        jobs = np.ascontiguousarray(jobs)
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        # a read-only job list is recognised by identity (the kernel object
        # keeps its lists that way), anything else by checksum
        jobs_id = ('id', id(jobs)) if not jobs.flags.writeable else \
            ('crc', zlib.crc32(jobs.view(np.uint8)))
        key = (_ids(dgraphs), len(jobs), jobs_id,
               zlib.crc32(starts.view(np.uint8)), C, fields, self.tables,
               ngrad, maximin, bool(nodal), bool(mfma),
               None if merge_map is None else tuple(sorted(merge_map.items())))
        hit = self._layouts.get(key)
        if hit is not None:
            self._layouts.move_to_end(key)
            return hit
        tic = timer.tic if timer else (lambda *_: None)
        toc = timer.toc if timer else (lambda *_: None)
        tic('calculating launch configuration')
        lay = Layout()
        lay.dgraphs = list(dgraphs)          # keeps the ids in `key` alive
        lay.jobs_host = jobs
        tic('  arena and label classes')
        lay.arena, lay.arena_buf, _ = self._arena(dgraphs, fields)
        toc('  arena and label classes')
        lay.tab_bytes = self._table_bytes(lay.arena)
        # (the nodal-gradient solvers evaluate the microkernels directly)
        lay.gtab = self._global_tables(lay.arena) and not ngrad
        if ngrad or maximin:
            lay.tab_bytes = 0
        tic('  solver variants and launch order')
        part = self._partition(dgraphs, jobs, C, lay.tab_bytes, lay.gtab,
                               oc_only=ngrad or maximin, merge_map=merge_map,
                               nodal=nodal, mfma=mfma)
        jobs, lay.used, lay.order_host, lay.launches = part
        toc('  solver variants and launch order')
        tic('  job list to the device')
        lay.n_jobs = len(jobs)
        lay.b_jobs = runtime.DeviceBuffer(max(jobs.nbytes, 8))
        lay.b_order = runtime.DeviceBuffer(max(lay.order_host.nbytes, 4))
        lay.b_starts = runtime.DeviceBuffer(max(starts.nbytes, 4))
        # jobs travel in launch order: the kernel reads jobs[t] directly
        sorted_jobs = part.jobs_sorted if part.jobs_sorted is not None \
            else np.ascontiguousarray(jobs[lay.order_host])
        lay.b_jobs.upload(sorted_jobs.view(np.uint32))
        lay.b_order.upload(lay.order_host)
        lay.b_starts.upload(starts)
        # uploads were issued on the null stream; solver launches may go to
        # non-blocking streams, which do not wait for it
        runtime.synchronize()
        toc('  job list to the device')
        self._layouts[key] = lay
        while len(self._layouts) > self.layout_cache_size:
            self._layouts.popitem(last=False)    # freed with its last plan
        toc('calculating launch configuration')
        return lay

    def prepare(self, graphs, node_kernel, edge_kernel, p, q, eps, ftol, gtol,
                jobs, starts, nX, nY, nJ, traits, timer=None, packed=False,
                gramian_ptr=None, gradient_ptr=None, ngrad=False,
                maximin=None, merge_map=None):
        """Upload graphs / jobs, generate + compile code, partition the jobs.
        Returns a Plan whose launches can be replayed.  `gramian_ptr` /
        `gradient_ptr` (device addresses) make the kernels write into
        caller-owned memory, e.g. the tensor handed to the all-gather.
        `ngrad`: nodal outputs with their finite-difference Jacobian in the
        same launch (owner-computes solvers only: raises NotOwnerComputes if
        some pair does not fit one).  `maximin`: dict(diag=, diag_grad=,
        node_starts=, ld=) of device addresses -- the launch writes the
        maximin distance, its hotspot (and with `ngrad` its gradient) per
        pair instead of the nodal matrix (`maximin_distance`).  `merge_map`:
        {variant index: variant index} launch merging decided on a larger job
        list that `jobs` is a shard of (instead of by this list's counts)."""
        tic = timer.tic if timer else (lambda *_: None)
        toc = timer.toc if timer else (lambda *_: None)
        runtime.ensure_device(self.device)
        node_kernel_in, edge_kernel_in = node_kernel, edge_kernel
        dgraphs, edge_kernel, C, fields = self._graphs_and_kernels(
            graphs, node_kernel, edge_kernel, traits, timer, ngrad)
        lay = self._layout(dgraphs, jobs, starts, C, fields, timer, ngrad,
                           maximin is not None, merge_map,
                           nodal=traits.nodal is not False,
                           mfma=self._label_blind(edge_kernel_in))
        tab = lay.tab_bytes > 0

        tic('code generation')
        nodal = traits.nodal is not False
        # the loaded modules and the argument block of a (code signature, set
        # of variants): a repeated call -- new hyperparameters, same code --
        # renders, hashes and looks up nothing
        mkey = (self._code_signature(node_kernel, edge_kernel, p, dgraphs, C,
                                     nodal, tab, lay.gtab, ngrad,
                                     maximin is not None), tuple(lay.used),
                tuple(self.hipcc_extra), self.tables,
                None if self.occupancy is None
                else tuple(sorted(self.occupancy.items())))
        hit = self._module_sets.get(mkey)
        if hit is None:
            sources = self._sources(lay.used, node_kernel, edge_kernel, p,
                                    dgraphs, C, nodal, tab, lay.gtab, ngrad,
                                    maximin is not None)
            toc('code generation')
            tic('JIT')
            missing = [s for s in sources.values()
                       if jit.cache_key(s, self.hipcc_extra)
                       not in self._modules]
            if len(missing) > 1:
                jit.compile_many(missing, self.hipcc_extra)
            modules = {k: self._module(s) for k, s in sources.items()}
            toc('JIT')
            hit = self._module_sets[mkey] = (
                modules, self._params_dtype(node_kernel, edge_kernel, p))
        else:
            toc('code generation')
        modules, pd = hit

        plan = Plan()
        plan.layout = lay
        plan.traits, plan.C, plan.n_jobs = traits, C, lay.n_jobs
        plan.nX, plan.nY, plan.nJ = int(nX), int(nY), int(nJ)
        plan.packed = packed
        plan.keep = (lay.arena, lay.arena_buf, dgraphs)
        plan.order_host = lay.order_host
        flags = 0
        if traits.nodal is True or traits.nodal == 'block':
            flags |= F_NODAL
        if traits.nodal == 'block':
            flags |= F_BLOCK
        if traits.diagonal:
            flags |= F_DIAGONAL
        if traits.symmetric:
            flags |= F_SYMMETRIC
        if traits.lmin == 1:
            flags |= F_LMIN1
        if packed:
            flags |= F_PACKED
        if maximin and maximin.get('reference_compat'):
            flags |= F_REFCOMPAT
        rsize = np.dtype(self.real).itemsize
        n_out = lay.n_jobs if packed else plan.nX * plan.nY
        plan.n_out = n_out
        plan.maximin = maximin is not None
        plan.ngrad = ngrad
        plan.n_grad = n_out * plan.nJ if (C == 2 or ngrad) else 0

        launches = []
        for G in lay.launches:
            L = dict(G)
            L['module'] = modules[L['k']]
            L['tab'] = L.get('tab', False) \
                if isinstance(L['variant'], OCVariant) \
                else tab and L['variant'] not in (GENERAL, STREAM, MFMA)
            L['fn'] = fn = L['module'].function(
                self.kernel_name(L['variant'], C, nodal, L['tab'], ngrad,
                                 maximin is not None))
            if L['variant'] == STREAM:
                # few pairs: several workgroups per pair, a cooperative
                # launch (mgk_stream.h).  M = what the chip holds at once over
                # the pairs; one workgroup per pair from half the chip up.
                if L['dynamic_lds'] > 64 * 1024:
                    runtime.set_max_dynamic_lds(fn, L['dynamic_lds'])
                # (workgroups the chip holds at once, by this build's own
                # arithmetic as well as by the runtime's: a grid beyond it
                # would wait for workgroups that wait for it)
                per_cu = max(1, min(
                    runtime.max_active_blocks(fn, L['threads'],
                                              L['dynamic_lds']),
                    LDS_LIMIT // (L['dynamic_lds'] + 1024),
                    2048 // L['threads']))
                resident = self.props.compute_units * per_cu
                parts = int(os.environ.get('GD_STREAM_PARTS', 0)) or \
                    stream_parts(L.get('costs'), L['count'], resident,
                                 2 * self.props.compute_units)
                parts = max(1, min(parts, resident))
                slots = int(min(L['count'], max(1, resident // parts)))
                if parts == 1:
                    slots = int(min(L['count'],
                                    2 * self.props.compute_units))
                L['parts'], L['cooperative'] = parts, parts > 1
                L['grid'] = slots * parts
                L['scratch_bytes'] = slots * L['per_wg'] * rsize
                L['sync_bytes'] = slots * 4 * (
                    16 + 2 * parts * 4 * (rsize // 4)) if parts > 1 else 0
            elif L['variant'] == GENERAL:
                L['grid'] = int(min(L['count'],
                                    2 * self.props.compute_units))
                L['scratch_bytes'] = L['grid'] * L['per_wg'] * rsize
            if L['dynamic_lds'] > 64 * 1024:
                runtime.set_max_dynamic_lds(fn, L['dynamic_lds'])
            launches.append(L)
        plan.launches = launches
        # Streams the launches are dealt onto (LaunchSet): three by default
        # (short launches fill the tails of long ones: +4-6 % over one, the
        # dense molecular set +10 % over two), TWO for the double value plans
        # of one-wave static layouts -- latency-bound kernels at two or three
        # waves per SIMD whose dominant launch loses more residency to a
        # third concurrent grid than its tail gains: 168.4-168.6 against
        # 166.1-166.3 M pairs/s on the headline, alternating on one box
        # (profiles/sessions.md r5_session29; float: equal; double value +
        # gradient: three, 64.5 against 63.7 M; the same plans run to
        # convergence, ftol 1e-13 and 26 iterations instead of 17: three, 126.6
        # against 125.7 M, r5_session36 -- hence the tolerance in the rule).
        # Round 6: the float value plans too -- equal on the full matrix, and on
        # a rank's share of it (354 graphs, five launches) 0.342 against 0.373 ms
        # per step (double: 0.447 against 0.455; one stream: 0.513 / 0.392;
        # scripts/fixed_cost_experiment.sh, profiles/r06_fixed_cost.log).
        plan.stream_hint = 2 if (
            C == 1 and not nodal
            and not ngrad and len(launches) > 2 and ftol >= 1e-10
            and all(isinstance(L['variant'], OCVariant) and L['variant'].L
                    for L in launches)) else None

        # per-call device buffers (outputs, scratch) from the grow-only pool
        b_out = self._buffer('gramian', n_out * rsize)
        b_grad = self._buffer('gradient', plan.n_grad * rsize) \
            if plan.n_grad else None
        b_iters = self._buffer('iters', 4 * lay.n_jobs) \
            if self.record_iterations else None
        # (launches run concurrently on several streams: every launch that
        # keeps CG vectors in global memory gets a region of its own)
        scratch_bytes = 0
        for L in launches:
            if L.get('scratch_bytes', 0):
                L['scratch_offset'] = scratch_bytes
                scratch_bytes += -(-L['scratch_bytes'] // 256) * 256
        b_scratch = self._buffer('scratch', scratch_bytes) \
            if scratch_bytes else None
        sync_bytes = max([L.get('sync_bytes', 0) for L in launches] + [0])
        # (a buffer of its OWN per plan: the kernels leave the barrier cells
        # clean at THIS plan's slot stride -- a pooled buffer re-used by a
        # plan of another M had its counters on leftover partial sums, and
        # the second evaluation of a backend dead-locked in its first barrier)
        b_sync = self._zeroed_buffer(sync_bytes) if sync_bytes else None
        # global microkernel tables of this evaluation: values (and, for the
        # gradient solvers, one plane per hyperparameter) per pair of classes
        b_tables = None
        if 'tables' in modules:
            c = lay.arena.classes
            planes = 1 + (len(node_kernel.gen_expr('x1', 'x2')[1]) +
                          len(edge_kernel.gen_expr('x1', 'x2')[1])
                          if C == 2 else 0)
            n_tab = (c['nv']**2 + c['ne']**2) * planes
            b_tables = self._buffer('tables', n_tab * rsize)
        b_hot = self._buffer('hotspot', 4 * n_out) if maximin else None
        plan.buffers = dict(jobs=lay.b_jobs, order=lay.b_order,
                            starts=lay.b_starts, gramian=b_out,
                            gradient=b_grad, iters=b_iters, tables=b_tables,
                            hotspot=b_hot)

        # kernel argument blocks
        base = np.zeros((), dtype=pd)
        base['arena'] = lay.arena_buf.ptr
        base['jobs'] = lay.b_jobs.ptr
        base['starts'] = lay.b_starts.ptr
        base['gramian'] = gramian_ptr if gramian_ptr else b_out.ptr
        base['gradient'] = gradient_ptr if gradient_ptr else (
            b_grad.ptr if b_grad is not None else 0)
        base['iters'] = b_iters.ptr if b_iters is not None else 0
        base['scratch'] = b_scratch.ptr if b_scratch is not None else 0
        base['tables'] = b_tables.ptr if b_tables is not None else 0
        if maximin:
            base['diag'], base['diag_grad'] = maximin['diag'], \
                maximin.get('diag_grad', 0)
            base['node_starts'], base['diag_ld'] = maximin['node_starts'], \
                maximin['ld']
            base['hotspot'] = b_hot.ptr
        base['nX'], base['nY'], base['nJ'] = plan.nX, plan.nY, plan.nJ
        base['flags'] = flags
        if tab or b_tables is not None:
            c = lay.arena.classes
            base['n_vclass'], base['n_eclass'] = c['nv'], c['ne']
            base['vrep'], base['erep'] = c['vrep'], c['erep']
        base['q'] = q
        base['q0'] = q
        base['eps'], base['ftol'], base['gtol'] = eps, ftol, gtol
        for field, obj in (('node_kernel', node_kernel),
                           ('edge_kernel', edge_kernel), ('p_start', p)):
            dt, val = pack_theta(obj, self.real)
            if val is not None:
                base[field] = val
        fd = None
        if ngrad:
            # perturbed hyperparameter sets exp(log(theta) +- eps) of the
            # central differences (reference: pack_state(diff_grid=True),
            # _backend_cuda.py:230-245)
            fd = np.zeros((), dtype=self._params_fd_dtype(
                node_kernel, edge_kernel, p))
            fd['q_plus'] = np.exp(np.log(q) + eps)
            fd['q_minus'] = np.exp(np.log(q) - eps)
            for which, kern in (('node', node_kernel), ('edge', edge_kernel)):
                theta = np.array(list(flatten(kern.theta)), dtype=float)
                for j_, t_ in enumerate(theta):
                    fd[which + '_theta'][j_] = t_
                    for s_, delta in enumerate((eps, -eps)):
                        k2 = copy.deepcopy(kern)
                        lt = np.log(theta)
                        lt[j_] += delta
                        k2.theta = fold_like(np.exp(lt), k2.theta)
                        _, val = pack_theta(k2, self.real)
                        if val is not None:
                            fd[which + '_diff'][2 * j_ + s_] = val
        for L in launches:
            a = base.copy()
            a['order'] = lay.b_order.ptr + 4 * L['offset']
            a['jobs'] = lay.b_jobs.ptr + 8 * L['offset']
            a['g_capacity'] = L['gcap']
            a['n_launch_jobs'] = L['count']
            a['order_offset'] = L['offset']
            a['u_capacity'] = L['ucap']
            if L.get('scratch_bytes', 0):
                a['scratch'] = b_scratch.ptr + L['scratch_offset']
            if L.get('parts', 1) > 1:
                a['parts'], a['sync'] = L['parts'], b_sync.ptr
                plan.sync = (b_sync, L['sync_bytes'])
            if L.get('dense'):
                a['flags'] |= F_DENSE
            if fd is not None:
                f_ = fd.copy()
                f_['base'] = a
                L['args'] = f_.tobytes()
            else:
                L['args'] = a.tobytes()
        # launches that every solver launch depends on: the table kernel
        plan.pre_launches = []
        if b_tables is not None:
            c = lay.arena.classes
            plan.pre_launches.append(dict(
                variant=TABLES, module=modules['tables'],
                fn=modules['tables'].function(self.kernel_name(TABLES, C)),
                grid=int(-(-(c['nv']**2 + c['ne']**2) // 256)), threads=256,
                args=base.tobytes(), dynamic_lds=0))
        plan.params_dtype = pd
        self.last_plan = plan
        return plan

    def launch(self, plan, stream=None, concurrent=None):
        """Enqueue every launch of `plan` (asynchronous).  With `concurrent`
        (default: the backend's setting) each solver variant goes to its own
        HIP stream so that the short launches fill the tails of the long
        ones; `synchronize()` / `collect()` wait for all of them."""
        concurrent = self.concurrent if concurrent is None else concurrent
        if stream is not None:
            for L in plan.pre_launches + plan.launches:
                runtime.launch(L['fn'], L['grid'], L['threads'], L['args'],
                               stream=stream, dynamic_lds=L['dynamic_lds'],
                           cooperative=L.get('cooperative', False))
            return
        if self._launch_set is None:
            self._launch_set = LaunchSet()
        self._launch_set.enqueue(
            plan, serial=not concurrent or len(plan.launches) < 2)

    def synchronize(self):
        runtime.synchronize()

    def collect(self, plan, gramian=None, gradient=None):
        """Copy results back.  A packed plan returns one value (and nJ
        gradient entries) per job, in job order."""
        runtime.synchronize()
        rs = np.dtype(self.real)
        sync = getattr(plan, 'sync', None)
        if sync is not None:
            # the streamed solver's grid barriers: a slot whose parts did not
            # all arrive was poisoned by the watchdog (mgk_stream.h) -- its
            # results are not solutions
            cells = np.empty(sync[1] // 4, dtype=np.uint32)
            sync[0].download(cells)
            L = [L for L in plan.launches if L.get('parts', 1) > 1][0]
            stride = len(cells) // max(L['grid'] // L['parts'], 1)
            if np.any(cells[2::stride][:L['grid'] // L['parts']]):
                sync[0].upload(np.zeros(sync[0].nbytes, dtype=np.uint8))
                runtime.synchronize()
                raise RuntimeError(
                    'streamed solver: a grid-wide barrier of a cooperative '
                    f'launch ({L["grid"]} workgroups, {L["parts"]} per pair) '
                    'timed out -- the grid was not co-resident; set '
                    'GD_STREAM_PARTS=1 for one workgroup per pair')

        def fetch(buf, n, dest):
            # straight into the caller's array when it can take the bytes
            if (isinstance(dest, np.ndarray) and dest.dtype == rs
                    and dest.size == n and dest.flags.c_contiguous
                    and dest.flags.writeable):
                buf.download(dest.reshape(-1))
                return dest.reshape(-1)
            out = np.empty(n, dtype=rs)
            buf.download(out)
            if dest is not None:
                dest[:] = out.reshape(dest.shape)
            return out

        out = fetch(plan.buffers['gramian'], plan.n_out, gramian)
        grad = None
        if plan.C == 2 or getattr(plan, 'ngrad', False):
            grad = fetch(plan.buffers['gradient'], plan.n_grad, gradient)
        return out, grad

    def iterations(self, plan):
        if plan.buffers['iters'] is None:
            raise RuntimeError('create the backend with '
                               'record_iterations=True')
        it = np.empty(plan.n_jobs, dtype=np.uint32)
        plan.buffers['iters'].download(it)
        return it

    # -- nodal gradients: central finite differences in log-theta ----------------------
    def _nodal_gradient(self, graphs, node_kernel, edge_kernel, p, q, eps,
                        ftol, gtol, jobs, starts, gramian, gradient, nX, nY,
                        nJ, traits, timer):
        """Value + Jacobian of nodal outputs, as the reference defines them
        (template.cu:226-418): d/dp analytic, d/dq, d/d(node theta),
        d/d(edge theta) by central differences of re-solves at
        exp(log(theta) +- eps).  The reference warm-starts those re-solves
        in-kernel with tolerance gtol; here each perturbed system is a fresh
        value launch (hyperparameters are kernel arguments, no recompilation),
        which converges to the same differences."""
        def solve(nk, ek, qq, lmin):
            tr = traits._replace(eval_gradient=False, lmin=lmin)
            plan = self.prepare(graphs, nk, ek, p, qq, eps, ftol, gtol, jobs,
                                starts, nX, nY, nJ, tr, None)
            self.launch(plan)
            out, _ = self.collect(plan)
            return out.astype(np.float64)

        value = solve(node_kernel, edge_kernel, q, traits.lmin)
        gramian[:] = value
        shape = (nX, nJ) if traits.diagonal else (nX, nY, nJ)
        J = np.zeros(shape, dtype=np.float64, order='F')
        K = value.reshape(shape[:-1], order='F')

        # d/dp (analytic): K_nodal * (dp1/p1 + dp2/p2)   (template.cu:258-284)
        def node_p(gs):
            pv, dpv = [], []
            for g in gs:
                order = np.argsort(np.asarray(g.nodes['!i']))
                a, b = p(g.nodes)
                a = np.asarray(a, dtype=float)[order]
                b = np.asarray(b, dtype=float)
                b = b[:, order] if b.size else np.zeros((0, len(order)))
                pv.append(a)
                dpv.append(b)
            return np.concatenate(pv), np.concatenate(dpv, axis=1)
        n_p = len(list(flatten(p.theta)))
        if traits.diagonal or traits.symmetric:
            px, dpx = node_p(graphs)
            py, dpy = px, dpx
        else:
            sizes = np.array([len(g.nodes) for g in graphs])
            split = int(np.searchsorted(np.cumsum(sizes), nX, side='left')) + 1
            px, dpx = node_p(graphs[:split])
            py, dpy = node_p(graphs[split:])
        for j in range(n_p):
            if traits.diagonal:
                J[:, j] = K * 2 * dpx[j] / px
            else:
                J[:, :, j] = K * ((dpx[j] / px)[:, None]
                                  + (dpy[j] / py)[None, :])
        col = n_p

        def central(plus, minus, denom):
            nonlocal col
            d = (solve(*plus, 0) - solve(*minus, 0)) / denom
            J[..., col] = d.reshape(shape[:-1], order='F')
            col += 1

        central((node_kernel, edge_kernel, float(np.exp(np.log(q) + eps))),
                (node_kernel, edge_kernel, float(np.exp(np.log(q) - eps))),
                2 * eps * q)
        for which, kern in (('node', node_kernel), ('edge', edge_kernel)):
            theta = np.array(list(flatten(kern.theta)), dtype=float)
            for j in range(len(theta)):
                pair = []
                for delta in (eps, -eps):
                    k2 = copy.deepcopy(kern)
                    t = np.log(theta)
                    t[j] += delta
                    k2.theta = fold_like(np.exp(t), k2.theta)
                    pair.append(k2)
                if which == 'node':
                    central((pair[0], edge_kernel, q),
                            (pair[1], edge_kernel, q), 2 * eps * theta[j])
                else:
                    central((node_kernel, pair[0], q),
                            (node_kernel, pair[1], q), 2 * eps * theta[j])
        assert col == nJ, (col, nJ)
        gradient[:] = J.ravel(order='F')

    # -- maximin graph distance, fused ------------------------------------------------
    def maximin_distance(self, graphs, node_kernel, edge_kernel, p, q, eps,
                         ftol, gtol, jobs, nX, nY, nJ, traits, timer=None,
                         reference_compat=False):
        """Maximin distances (+ hotspots, + gradients with
        `traits.eval_gradient`) of the graph pairs in `jobs` without the
        nodal Gram matrix ever leaving the compute units (reference:
        metric/maximin/_backend.cu:40-407, one fused CUDA kernel): a `diag`
        launch leaves the nodal self-similarities of every graph (and their
        Jacobian) on the device, the pair launch reduces row minima / column
        minima / their maximum in LDS (mgk_oc.h, MAXIMIN).  `traits` are
        graph-level (`nodal=False`); nX, nY count graphs.  Owner-computes
        solvers only: raises NotOwnerComputes otherwise.  `reference_compat`:
        the gradient columns from q on are formed with k12 (and the distance in
        the denominator) of the last perturbed solve, as the reference's kernel
        does (_backend.cu:383); default: the unperturbed solution.  Returns (distance
        [nX nY], hotspot int32 [nX nY], gradient [nX nY nJ] or None), flat
        column-major like the other outputs."""
        grad = traits.eval_gradient is True
        n = len(graphs)
        sizes = np.array([len(g.nodes) for g in graphs], dtype=np.uint32)
        node_starts = np.zeros(n + 1, dtype=np.uint32)
        np.cumsum(sizes, out=node_starts[1:])
        total = int(node_starts[-1])
        rs = np.dtype(self.real)
        # 1. nodal self-similarities (diag, nodal): stays on the device
        djobs = np.zeros(n, dtype=np.dtype([('i', np.uint32),
                                            ('j', np.uint32)]))
        djobs['i'] = djobs['j'] = np.arange(n)
        dtraits = traits._replace(symmetric=False, nodal=True, diagonal=True,
                                  eval_gradient=grad)
        dplan = self.prepare(graphs, node_kernel, edge_kernel, p, q, eps,
                             ftol, gtol, djobs, node_starts, total, 1, nJ,
                             dtraits, timer, ngrad=grad)
        if any(not isinstance(L['variant'], OCVariant)
               for L in dplan.launches):
            raise NotOwnerComputes
        self.launch(dplan)
        b_diag = self._buffer('mm_diag', total * rs.itemsize)
        b_dgrad = self._buffer('mm_diag_grad', total * nJ * rs.itemsize) \
            if grad else None
        b_ns = self._buffer('mm_node_starts', node_starts.nbytes)
        runtime.synchronize()
        runtime.check(runtime.lib().gd_memcpy_d2d(
            b_diag.ptr, dplan.buffers['gramian'].ptr, total * rs.itemsize,
            None))
        if grad:
            runtime.check(runtime.lib().gd_memcpy_d2d(
                b_dgrad.ptr, dplan.buffers['gradient'].ptr,
                total * nJ * rs.itemsize, None))
        b_ns.upload(node_starts)
        runtime.synchronize()
        # 2. the pairs
        starts = np.arange(n + 1, dtype=np.uint32)
        if not traits.symmetric:
            starts[nX:] = np.arange(n - nX + 1)
        mtraits = traits._replace(nodal=True, eval_gradient=grad)
        plan = self.prepare(
            graphs, node_kernel, edge_kernel, p, q, eps, ftol, gtol, jobs,
            starts, nX, nY, nJ, mtraits, timer, ngrad=grad,
            maximin=dict(diag=b_diag.ptr,
                         diag_grad=b_dgrad.ptr if grad else 0,
                         node_starts=b_ns.ptr, ld=total,
                         reference_compat=bool(reference_compat)))
        if grad:
            plan.buffers['gradient'].zero()
        self.launch(plan)
        dist, g = self.collect(plan)
        hot = np.empty(plan.n_out, dtype=np.int32)
        plan.buffers['hotspot'].download(hot)
        return dist, hot, g

    # -- the reference's backend call ---------------------------------------------------
    def __call__(self, graphs, node_kernel, edge_kernel, p, q, eps, ftol,
                 gtol, jobs, starts, gramian, gradient, nX, nY, nJ, traits,
                 timer):
        if traits.eval_gradient is True and traits.nodal is True:
            # nodal Jacobian: in the same launch as the solve (owner-computes
            # solvers, mgk_oc.h NGRAD); graphs they do not cover take the
            # host-orchestrated re-launches
            try:
                plan = self.prepare(graphs, node_kernel, edge_kernel, p, q,
                                    eps, ftol, gtol, jobs, starts, nX, nY,
                                    nJ, traits, timer, ngrad=True) \
                    if self.nodal_gradient_in_kernel else None
            except NotOwnerComputes:
                plan = None
            timer.tic('GPU kernel execution')
            if plan is not None:
                self.launch(plan)
                runtime.synchronize()
                timer.toc('GPU kernel execution')
                self.collect(plan, gramian, gradient)
                return
            self._nodal_gradient(graphs, node_kernel, edge_kernel, p, q, eps,
                                 ftol, gtol, jobs, starts, gramian, gradient,
                                 nX, nY, nJ, traits, timer)
            timer.toc('GPU kernel execution')
            return
        if traits.eval_gradient is True and traits.nodal == 'block':
            # the reference computes no gradient in this mode either
            # (template.cu:226,422: neither branch is compiled for 'block')
            traits = traits._replace(eval_gradient=False)
            gradient = None
        # (the host side of a first call makes a few thousand small objects --
        # packed-graph handles, cookies -- none of them part of a cycle: the
        # cyclic collector, which would walk the caller's whole heap of graphs
        # for ~8 ms if its threshold fell inside, waits until the call is over)
        # (counted and thread-safe, GD_PAUSE_GC=0 turns it off: util/gcpause.py)
        with gcpause.paused():
            plan = self.prepare(graphs, node_kernel, edge_kernel, p, q, eps,
                                ftol, gtol, jobs, starts, nX, nY, nJ, traits,
                                timer)
            timer.tic('GPU kernel execution')
            self.launch(plan)
            runtime.synchronize()
            timer.toc('GPU kernel execution')
            self.collect(plan, gramian, gradient)
