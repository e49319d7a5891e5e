// Streamed solver for LARGE graph pairs on MI355X (gfx950, wave64): product
// graphs of 1e4 ... 1e6 rows whose CG vectors fit neither registers nor LDS
// -- the regime of the reference's protein benchmark
// (example/perfbench/protein-time-to-solution.py:1-58; solver scratch:
// graphdot/kernel/marginalized/_scratch.py:20-36, marginalized_kernel.h:20-38).
//
// Same system, iteration and stopping rules as every other solver here
// (reference: graphdot/cpp/marginalized_kernel.h:394-461); what differs from
// `general_solver` (mgk_solver.h), which it replaces for value solves:
//
//  * ONE stage instead of two.  The owner of row (iA, iB) sums its whole
//    Kronecker double sum  sum_{a in adj(iA)} sum_{b in adj(iB)} E[a,b]
//    p[jA(a), jB(b)]  -- no U[a, iB] array through global memory (written
//    once and read twice per iteration by general_solver).
//  * One graph of the pair (B: the smaller image) is staged into LDS once per
//    pair -- row pointers, column indices and edge records: every per-term
//    operand of B is a ds_read.  general_solver read them from global memory
//    in a dependent chain per term.
//  * The other graph (A) is STREAMED row by row: a workgroup step takes G
//    rows iA (G groups of LB = ceil64(nB) lanes; lane lb of group g owns row
//    (iA = base + g, iB = lb)), and for A_ROWS neighbours a of iA at a time
//    the rows p[jA(a), :] are copied -- coalesced -- from the global vector
//    into LDS; the gathers p[jA(a), jB(b)] of all lanes then hit that copy.
//    Graph A's edge record and column are the same for all lanes of a wave:
//    scalar operands.
//  * The Jacobi diagonal is evaluated once per pair into the scratch
//    (general_solver: four evaluations of the node microkernel and four
//    divisions per row and iteration), A p is kept from the mat-vec for the
//    update (general_solver re-summed U), and every vector pass is flat and
//    coalesced.
//
// Scratch per pair slot: [x | r | p | Ap | diag (| second solution)], N reals each
// (prm.u_capacity reals).  Dynamic LDS: [image of B | A_ROWS rows of p per
// group | one partial sum per lane], sized pair by pair.  The edge microkernel is evaluated per term and iteration, as the
// reference does (marginalized_kernel.h:299-300,346): nnz_A nnz_B values do
// not fit anywhere (5e7 per pair), and streaming them from HBM costs what
// re-evaluating them does.
#ifndef GRAPHDOT_HIP_MGK_STREAM_H_
#define GRAPHDOT_HIP_MGK_STREAM_H_
#include "mgk_solver.h"

namespace graphdot {
namespace mgk {

template<class real, int TPB, int C, class Graph, class NodeK, class EdgeK, class PStart>
struct stream_solver {
    static_assert(C == 1 || C == 2, "C = 1: values; C = 2: values and the analytic gradient");
    // C = 2: the two right-hand sides of the reference's compute_duo
    // (marginalized_kernel.h:492-804), D q^2/q0^2 and p1 (x) p2, are solved ONE
    // AFTER THE OTHER by the same iteration (as the static double solvers of
    // mgk_oc.h do, SEQ): the stacked system's stopping rule rTr_0 + rTr_1 <
    // (1e-10 * 2N)^2 (:769) as half the budget for the first system and what
    // it left for the second.  Then the analytic derivative (:806-997): a flat
    // pass over the rows for d/dp, d/dq and the node hyperparameters, and for
    // the edge hyperparameters one more streamed walk over the terms -- the
    // rows of the FIRST solution staged where the rows of p were, the
    // microkernel's Jacobian in place of its value, weighted with the second
    // solution at the row.
    constexpr static int n_jac = PStart::jac_dims + 1 + NodeK::jac_dims + EdgeK::jac_dims;
    constexpr static int off_q = PStart::jac_dims;
    constexpr static int off_v = off_q + 1;
    constexpr static int off_e = off_v + NodeK::jac_dims;
    constexpr static int NVEC = C == 2 ? 6 : 5;      // vectors of N reals in the scratch
    using P = params_t<real, Graph, NodeK, EdgeK, PStart>;
    using node_t = typename Graph::node_t;
    using edge_t = typename Graph::edge_t;
    constexpr static int W = TPB / 64;
    // rows of p staged per pass and group.  A pass costs every lane the reads
    // of its segment's edge records and columns whatever the number of staged
    // rows: 2 / 4 / 8 / 12 rows: 173 / 136 / 122 / 127 ms on the 528 pairs of
    // bench.py --config large in float (12: the records of the staged rows
    // spill).  Measured beside it and NOT kept (profiles/sessions.md
    // r5_session20): the microkernel on two staged rows at once with the
    // packed float instructions -- half the vector instructions per term, the
    // same time --, and the loads of the next pass's rows issued in front of
    // the products of this one, two stage buffers, one barrier per pass --
    // 137 against 136 ms: the passes wait neither for the vector pipe nor for
    // those loads.  What did pay: the staged values of a COLUMN side by side
    // (`load_staged`: one 16-byte read per four terms instead of four gathers)
    // -- 122 -> 100.6 ms, 136 pairs 63.7 -> 43.7 ms, one pair 1.13 -> 0.87 ms,
    // value + gradient of 36 pairs 60.5 -> 40.6 ms.  Double: 8 rows too (316
    // -> 303 ms on the 528 pairs, although more of B's images then stay in L2).
    constexpr static int A_ROWS = 8;                  // (HIPBackend STREAM_ROWS)

    constexpr static int SEG_CAP = 16;                    // neighbours of B per lane segment (doubled until the segments fit)
    constexpr static int MAX_LAYERS = 64;             // nB <= TPB = 1024 = 64 * 16
    constexpr static unsigned LDS_BUDGET = 159 * 1024;   // dynamic LDS a pair may ask for (HIPBackend.stream_lds_bytes)

    struct lds_t {
        real red[2 * W];
        int lay_off[MAX_LAYERS + 1];
    };

    // bytes of the contiguous image [degree .. perm] of a graph, in 16-byte units
    __device__ static __forceinline__ unsigned image_words(graph_header_t const &h) {
        return (h.perm + 2u * (unsigned)h.n_node - h.degree + 15u) / 16u;
    }

    // the A_ROWS staged values of one column of p (A_ROWS reals, 16-byte
    // aligned: the stage starts on a 16-byte boundary and a column is 32 bytes)
    static_assert((A_ROWS * sizeof(real)) % 16 == 0, "a staged column is read in 16-byte pieces");
    __device__ static __forceinline__ void load_staged(real const *src, real (&v)[A_ROWS]) {
        typedef unsigned v4 __attribute__((ext_vector_type(4)));
        constexpr int NQ = A_ROWS * sizeof(real) / 16;
        v4 q[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = reinterpret_cast<v4 const *>(src)[k];
        __builtin_memcpy(v, q, sizeof(v));
    }

    template<class V> __device__ static __forceinline__ V pick(bool second, V const &a, V const &b) {
        return second ? b : a;
    }

    // ---- several workgroups per pair ---------------------------------------
    // A launch of few pairs (one protein against itself: the reference's
    // protein-time-to-solution.py) would leave all but a few compute units
    // idle: M = prm.parts workgroups then share a pair.  Part m owns the rows
    // iA in [nA m / M, nA (m + 1) / M): their share of the mat-vec, of every
    // vector pass and of the scalar products; p is read by all.  Three
    // grid-wide barriers per iteration (p complete | pAp | rTr, rTz), through
    // device memory: the launch is COOPERATIVE (gd_launch_cooperative: every
    // workgroup resident, and the device runs one such kernel at a time), the
    // barrier a counter and a generation number per pair slot -- the last
    // part to arrive resets the counter and bumps the generation, so the
    // cells are clean when the kernel ends and need no host-side reset.
    // Scalar products: every part publishes its partial sums, all parts add
    // the M partials up in part order -- the same numbers, hence the same
    // control flow, in every part.  M = 1: none of this is executed.
    constexpr static unsigned SYNC_HEAD = 16;        // words: [count, generation, poisoned, pad]
    __device__ static __forceinline__ unsigned sync_words(unsigned M) {
        return SYNC_HEAD + 2u * M * 4u * (unsigned)(sizeof(real) / 4u);
    }
    struct group_t {
        unsigned M, part;
        unsigned *cells;          // [count, generation]
        real *partial;            // [2][M][4]
        unsigned gen, epoch;
        __device__ __forceinline__ void barrier() {
            if (M == 1) {
                __syncthreads();
                return;
            }
            __syncthreads();      // (this workgroup's stores are issued)
            if (threadIdx.x == 0) {
                __threadfence();  // release: they are visible device-wide
                const unsigned arrived = atomicAdd(&cells[0], 1u);
                if (arrived == M - 1) {
                    atomicExch(&cells[0], 0u);
                    __threadfence();
                    atomicAdd(&cells[1], 1u);
                } else {
                    // (a part that never arrives -- a grid that is not fully
                    // resident after all, a diverged part -- must not hang
                    // the device: after ~2 s the slot is POISONED, cells[2],
                    // every barrier of it falls through, and the host raises
                    // when it collects the results)
                    unsigned spins = 0;
                    while (__hip_atomic_load(&cells[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                        __builtin_amdgcn_s_sleep(2);
                        if ((++spins & 0xFFFu) == 0u &&
                            (spins > (1u << 24) ||
                             __hip_atomic_load(&cells[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            atomicExch(&cells[2], 1u);
                            break;
                        }
                    }
                }
                __threadfence();  // acquire: the other parts' stores are read afresh
            }
            ++gen;
            __syncthreads();
        }
        // v[k], k < NV: block-level sums (the same in every thread) -> sums over the parts
        template<int NV> __device__ __forceinline__ void sum(real (&v)[NV], real *red) {
            if (M == 1) return;
            real *const mine = partial + ((size_t)(epoch & 1u) * M + part) * 4u;
            if (threadIdx.x == 0)
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    __hip_atomic_store(&mine[k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            barrier();
            if (threadIdx.x < 64) {
                real const *const all = partial + (size_t)(epoch & 1u) * M * 4u;
                real a[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) a[k] = 0;
                for (unsigned m = threadIdx.x; m < M; m += 64)
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        a[k] += __hip_atomic_load(&all[(size_t)m * 4u + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < NV; ++k) a[k] = wave::sum(a[k]);
                if (threadIdx.x == 0)
#pragma unroll
                    for (int k = 0; k < NV; ++k) red[k] = a[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = red[k];
            __syncthreads();
            ++epoch;
        }
    };

    __device__ static __forceinline__ void run(P const &prm, lds_t &lds, char *dyn, real *scratch_all) {
        const int tid = threadIdx.x;
        real *const red = lds.red;
        int *const lay_off = lds.lay_off;
        group_t grp;
        grp.M = prm.parts > 1u ? prm.parts : 1u;
        const unsigned slot = blockIdx.x / grp.M, n_slots = gridDim.x / grp.M;
        grp.part = blockIdx.x - slot * grp.M;
        grp.cells = prm.sync + (size_t)slot * sync_words(grp.M);
        grp.partial = reinterpret_cast<real *>(grp.cells + SYNC_HEAD);
        grp.epoch = 0;
        grp.gen = grp.M > 1u ? __hip_atomic_load(&grp.cells[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        real *const scratch = scratch_all + (size_t)slot * prm.u_capacity;
        graph_header_t const *const headers = reinterpret_cast<graph_header_t const *>(prm.arena);
        char *const lG = dyn;

        for (unsigned t = slot; t < prm.n_launch_jobs; t += n_slots) {
            const job_t job = prm.jobs[t];
            const graph_header_t h1 = headers[job.i], h2 = headers[job.j];
            // B, the LDS-resident graph: the smaller image among the graphs of
            // at most TPB nodes (ties: graph 2); the staged rows of p lie
            // behind it -- HIPBackend.stream_lds_bytes sizes the launch's
            // region by the same rule, pair by pair
            const unsigned w1 = image_words(h1), w2 = image_words(h2);
            const bool ok1 = h1.n_node <= TPB, ok2 = h2.n_node <= TPB;
            const bool sw_ = ok1 && (!ok2 || w1 < w2);     // B = graph 1
            const graph_header_t hA = sw_ ? h2 : h1, hB = sw_ ? h1 : h2;
            const unsigned wB = sw_ ? w1 : w2;
            // Does B's image fit the LDS beside the staged rows (at their
            // largest: A_ROWS + 1 reals per lane)?  If not -- the double
            // build's largest graphs, 16-byte edge records -- B is read
            // where it lies, from L2: slower per term, but the pair keeps
            // the streamed solver and its M workgroups.  The pair's body is
            // instantiated for either case (`in_lds`): a view of B through
            // one pointer type would turn every read into a flat load.
            const bool fits = wB * 16u + (unsigned)((A_ROWS + 1) * TPB * sizeof(real)) <= LDS_BUDGET;
            auto solve_pair = [&](auto in_lds, auto swapped) {
            constexpr bool B_IN_LDS = decltype(in_lds)::value;
            // (which graph is B, at compile time: as a run-time flag the
            // argument order of every microkernel call was two selects per
            // leaf of both records, per term -- four of the ten vector
            // instructions of a term of the benchmark's edge kernel; 140.9 ->
            // 136.0 ms on the 528 pairs of bench.py --config large)
            constexpr bool sw = decltype(swapped)::value;
            const Graph gA(prm.arena, hA);
            const Graph gB = [&] {
                if constexpr (B_IN_LDS) return Graph(lG - hB.degree, hB);
                else return Graph(prm.arena, hB);
            }();
            real *const stage = reinterpret_cast<real *>(lG + (B_IN_LDS ? wB * 16u : 0u));
            const int nA = hA.n_node, nB = hB.n_node, N = nA * nB;
            const real q = prm.q, q0 = prm.q0;
            const real inv1q2 = real(1) / ((real(1) - q) * (real(1) - q));
            const real bscale = q * q / (q0 * q0);
            real *const X = scratch;                 // first solution ("YDq")
            real *const Rv = X + (size_t)N;
            real *const Pv = Rv + (size_t)N;
            real *const AP = Pv + (size_t)N;
            real *const DG = AP + (size_t)N;
            [[maybe_unused]] real *const X2 = DG + (size_t)N;     // C = 2: second solution ("Yp")

            grp.barrier();       // the previous pair is done with LDS and scratch
            // this part's rows of A (and of every vector)
            const int rlo = (int)((long)nA * grp.part / grp.M), rhi = (int)((long)nA * (grp.part + 1u) / grp.M);
            const int ilo = rlo * nB, ihi = rhi * nB;
            if constexpr (B_IN_LDS) {
                typedef unsigned v4 __attribute__((ext_vector_type(4)));
                const v4 *const src = reinterpret_cast<const v4 *>(prm.arena + hB.degree);
                v4 *const dst = reinterpret_cast<v4 *>(lG);
                for (unsigned w = tid; w < wB; w += TPB) dst[w] = src[w];
            }
            __syncthreads();

            // ---- virtual rows of B ----------------------------------------------
            // A lane that owned a whole node of B would walk its d_B neighbours
            // for every neighbour a of iA, and the workgroup moves in lockstep
            // (the staged rows of p are shared): with 3 ... 35 neighbours per
            // node every wave waits for the one that holds the hubs -- the
            // step takes max(d_B) where the mean is a third of it.  So a node
            // is cut into SEGMENTS of at most CAP neighbours and a lane owns a
            // segment.  Nodes are stored by descending degree, so layer l --
            // the (l + 1)-th segments of the nodes with more than l CAP
            // neighbours -- is a prefix 0 .. c_l - 1 of the node list: segment
            // s of layer l belongs to node s - off_l, no table.  Layers are
            // laid out one after the other (each sorted by length); the
            // partial sums meet in LDS, the lane of a node's first segment
            // adds them up in layer order (deterministic).  CAP: 16, doubled
            // until the segments fit the workgroup's lanes.
            int cap = SEG_CAP, nV = 0, nlay = 0;
            for (;; cap *= 2) {
                // c_l by binary search in the degree-sorted node list (every
                // thread the same numbers: wave-uniform control flow)
                nV = 0;
                nlay = 0;
                for (int l = 0; l < MAX_LAYERS; ++l) {
                    int c = nB;
                    if (l > 0) {
                        int lo = 0, hi = nB;       // first node with degree <= l cap
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            const int d = (int)gB.rowptr[mid + 1] - (int)gB.rowptr[mid];
                            if (d > l * cap) lo = mid + 1;
                            else hi = mid;
                        }
                        c = lo;
                    }
                    if (c == 0) break;
                    if (tid == 0) lay_off[l] = nV;
                    nV += c;
                    nlay = l + 1;
                }
                if (nV <= TPB) break;
            }
            if (tid == 0) lay_off[nlay] = nV;
            __syncthreads();
            // thread -> (group g, virtual row lb of the group); G rows of A per step
            const int LB = (nV + 63) & ~63;
            const int G = TPB / LB;              // >= 1: nV <= TPB
            const int g = tid / LB, lb = tid - g * LB;
            const bool seg_ok = g < G && lb < nV;          // owns a segment
            const bool lane_ok = g < G && lb < nB;         // owns a node (its first segment)
            int lay = 0;
            for (int l = 1; l < nlay; ++l) lay += lb >= lay_off[l] ? 1 : 0;
            const int iBs = seg_ok ? lb - lay_off[lay] : 0;  // node of this lane's segment
            const int seg_b0 = seg_ok ? (int)gB.rowptr[iBs] + lay * cap : 0;
            const int seg_end = seg_ok ? (int)gB.rowptr[iBs + 1] : 0;
            const int seg_b1 = seg_b0 + cap < seg_end ? seg_b0 + cap : seg_end;
            real *const st = stage + (size_t)g * A_ROWS * nB;
            real *const ys = stage + (size_t)G * A_ROWS * nB + (size_t)g * LB;

            unsigned it_total = 0;
            real rTr_first = 0;
            for (int sys = 0; sys < C; ++sys) {
            real *const Xs = sys == 0 ? X : X2;
            // ---- diagonal, right-hand side, start vectors ---------------------
            real rTz = 0;
            for (int i = ilo + tid; i < ihi; i += TPB) {
                const int iA = i / nB, iB = i - iA * nB;
                const real dx = real(gA.degree[iA]) * real(gB.degree[iB]) * inv1q2;
                const node_t vA = gA.node[iA], vB = gB.node[iB];
                const real vx = real(prm.node_kernel(pick(sw, vA, vB), pick(sw, vB, vA)));
                const real b = sys == 0 ? dx * bscale : real(prm.p_start(vA)) * real(prm.p_start(vB));
                const real mi = vx / dx;
                DG[i] = dx / vx;
                Xs[i] = 0;
                Rv[i] = b;
                Pv[i] = b * mi;
                rTz += b * b * mi;
            }
            rTz = block_reduce<real, W>::sum(rTz, red);
            {
                real v[1] = {rTz};
                grp.sum(v, red);
                rTz = v[0];
            }

            const real tol = C == 2 ? real(1e-10) * real(2 * N) : prm.ftol * real(N);
            const real tol2 = C == 2 ? (sys == 0 ? tol * tol * real(0.5) : tol * tol - rTr_first) : tol * tol;
            unsigned it = 0;
            for (; it < (unsigned)N && rTz != real(0); ++it) {
                grp.barrier();       // p of this iteration is in memory, every part's rows of it
                real pAp = 0;
                for (int base = rlo; base < rhi; base += G) {
                    const int iA = base + g;
                    const bool row_ok = g < G && iA < rhi;
                    const int rsA = row_ok ? (int)gA.rowptr[iA] : 0;
                    const int dA = row_ok ? (int)gA.rowptr[iA + 1] - rsA : 0;
                    // longest row of the step (workgroup-uniform)
                    int dmax = 0;
                    for (int k = 0; k < G && base + k < rhi; ++k) {
                        const int d = (int)gA.rowptr[base + k + 1] - (int)gA.rowptr[base + k];
                        dmax = d > dmax ? d : dmax;
                    }
                    const int b0 = seg_b0;
                    const int b1 = row_ok ? seg_b1 : b0;
                    real acc = 0;
                    __syncthreads();     // the partial sums of the step before are read
                    for (int a0 = 0; a0 < dmax; a0 += A_ROWS) {
                        __syncthreads();     // the rows staged before are consumed
                        // (wave-uniform -- the lanes of a wave share their row of A --
                        // and said so: the tests `u < nv` below are scalar branches
                        // then, not an exec mask saved and restored around every term;
                        // may be <= 0)
                        const int nv = __builtin_amdgcn_readfirstlane(dA - a0 < A_ROWS ? dA - a0 : A_ROWS);
                        edge_t eA[A_ROWS];
#pragma unroll
                        for (int u = 0; u < A_ROWS; ++u) {
                            if (u < nv) {
                                eA[u] = gA.edge[rsA + a0 + u];
                                const unsigned jA = gA.nz[rsA + a0 + u].j;
                                if (lane_ok) st[lb * A_ROWS + u] = Pv[(size_t)jA * nB + lb];
                            }
                        }
                        __syncthreads();
                        // A pass of nv rows: the terms of its FULL groups of four rows
                        // are computed without tests between them -- four independent
                        // chains the scheduler interleaves (and packs: the weights and
                        // staged values as v_pk_mul / v_pk_fma_f32, B's weight factored
                        // out of the sum) --, the rows of the last, partial group behind
                        // scalar tests.  (All eight rows behind tests: 100.4 ms on the
                        // 528 pairs of bench.py --config large; full passes of eight
                        // without: 92.0; groups of four: 89.7.)
                        auto pass = [&](auto full_groups) {
                            constexpr int NF = 4 * decltype(full_groups)::value;    // rows without a test
                            for (int b = b0; b < b1; ++b) {
                                const edge_t eB = gB.edge[b];
                                const unsigned col = gB.nz[b].j;
                                // (the A_ROWS staged values of a column lie side by
                                // side: one or two 16-byte reads instead of A_ROWS
                                // gathers of 4 or 8 bytes)
                                real pv[A_ROWS];
                                load_staged(st + (size_t)col * A_ROWS, pv);
                                if constexpr (NF > 0) {
                                    real part[NF];
#pragma unroll
                                    for (int u = 0; u < NF; ++u)
                                        part[u] = real(prm.edge_kernel(pick(sw, eA[u], eB), pick(sw, eB, eA[u]))) * pv[u];
#pragma unroll
                                    for (int u = 0; u < NF; ++u) acc += part[u];
                                }
#pragma unroll
                                for (int u = NF; u < A_ROWS && u < NF + 4; ++u) {
                                    if (u < nv) {
                                        const real e = real(prm.edge_kernel(pick(sw, eA[u], eB), pick(sw, eB, eA[u])));
                                        acc += e * pv[u];
                                    }
                                }
                            }
                        };
                        static_assert(A_ROWS == 4 || A_ROWS == 8, "passes of one or two groups of four rows");
                        if (nv >= 8 && A_ROWS == 8) pass(std::integral_constant<int, A_ROWS == 8 ? 2 : 1>{});
                        else if (nv >= 4) pass(std::integral_constant<int, 1>{});
                        else pass(std::integral_constant<int, 0>{});
                    }
                    // partial sums of the segments -> their nodes
                    if (seg_ok) ys[lb] = acc;
                    __syncthreads();
                    if (lane_ok && row_ok) {
                        for (int l = 1; l < nlay; ++l)
                            if (lb < lay_off[l + 1] - lay_off[l]) acc += ys[lay_off[l] + lb];
                        const size_t i = (size_t)iA * nB + lb;
                        const real pv = Pv[i];
                        const real Ap = DG[i] * pv - acc;
                        AP[i] = Ap;
                        pAp += pv * Ap;
                    }
                }
                pAp = block_reduce<real, W>::sum(pAp, red);    // (its barriers publish AP)
                {
                    real v[1] = {pAp};
                    grp.sum(v, red);
                    pAp = v[0];
                }
                // (pAp != pAp: a NaN must end the solve, not run it for N
                // iterations -- the reference's rules do not stop on one)
                if (pAp == real(0) || pAp != pAp) break;
                const real alpha = rTz / pAp;
                real rTr = 0, rTz_next = 0;
                for (int i = ilo + tid; i < ihi; i += TPB) {
                    Xs[i] += alpha * Pv[i];
                    const real rv = Rv[i] - alpha * AP[i];
                    Rv[i] = rv;
                    rTr += rv * rv;
                    rTz_next += rv * rv / DG[i];
                }
                block_reduce<real, W>::sum2(rTr, rTz_next, red);
                {
                    real v[2] = {rTr, rTz_next};
                    grp.sum(v, red);
                    rTr = v[0];
                    rTz_next = v[1];
                }
                if (sys == 0) rTr_first = rTr;
                if (rTr < tol2) {
                    ++it;
                    break;
                }
                const real beta = rTz_next / rTz;
                for (int i = ilo + tid; i < ihi; i += TPB) Pv[i] = Rv[i] / DG[i] + beta * Pv[i];
                rTz = rTz_next;
            }
            it_total += it;
            }       // sys
            const unsigned it = C == 2 ? (it_total + 1u) / 2u : it_total;     // (iterations per system)
            __syncthreads();
            if (prm.iters != nullptr && tid == 0 && grp.part == 0) prm.iters[prm.order[t]] = it;

            // ---- output (conventions of pair_solver / template.cu:100-224) ----
            const unsigned flags = prm.flags;
            const unsigned I1 = prm.starts[job.i], I2 = prm.starts[job.j];
            const bool mirror = (flags & F_SYMMETRIC) && job.i != job.j;
            const int n2 = h2.n_node;
            real ksum = 0;
            for (int i = ilo + tid; i < ihi; i += TPB) {
                const int iA = i / nB, iB = i - iA * nB;
                const node_t vA = gA.node[iA], vB = gB.node[iB];
                real xi = X[i];
                if (flags & F_LMIN1)
                    xi -= real(prm.node_kernel(pick(sw, vA, vB), pick(sw, vB, vA))) * bscale;
                const real rv = xi * real(prm.p_start(vA)) * real(prm.p_start(vB));
                ksum += rv;
                if (flags & F_NODAL) {
                    const unsigned oA = gA.perm[iA], oB = gB.perm[iB];
                    const unsigned o1 = sw ? oB : oA, o2 = sw ? oA : oB;
                    if (flags & F_BLOCK) {
                        prm.gramian[I1 + o1 + o2 * n2] = rv;
                    } else if (flags & F_DIAGONAL) {
                        if (o1 == o2) prm.gramian[I1 + o1] = rv;
                    } else {
                        prm.gramian[(size_t)(I1 + o1) + (size_t)prm.nX * (I2 + o2)] = rv;
                        if (mirror) prm.gramian[(size_t)(I2 + o2) + (size_t)prm.nX * (I1 + o1)] = rv;
                    }
                }
            }
            if (!(flags & F_NODAL)) {
                ksum = block_reduce<real, W>::sum(ksum, red);
                {
                    real v[1] = {ksum};
                    grp.sum(v, red);
                    ksum = v[0];
                }
                if (tid == 0 && grp.part == 0) {
                    if (flags & F_PACKED) {
                        prm.gramian[prm.order[t]] = ksum;
                    } else if (flags & F_DIAGONAL) {
                        prm.gramian[I1] = ksum;
                    } else {
                        prm.gramian[(size_t)I1 + (size_t)prm.nX * I2] = ksum;
                        if (mirror) prm.gramian[(size_t)I2 + (size_t)prm.nX * I1] = ksum;
                    }
                }
            }

            if constexpr (C == 2) {
                // ---- analytic derivative (marginalized_kernel.h:831-994) --------
                grp.barrier();       // both solutions complete, every part's rows
                real jac[n_jac];
#pragma unroll
                for (int j = 0; j < n_jac; ++j) jac[j] = 0;
                const real Q = real(1) / (real(1) - q), Q3 = Q * Q * Q;
                for (int i = ilo + tid; i < ihi; i += TPB) {
                    const int iA = i / nB, iB = i - iA * nB;
                    const node_t vA = gA.node[iA], vB = gB.node[iB];
                    const node_t v1 = pick(sw, vA, vB), v2 = pick(sw, vB, vA);
                    const real p1 = prm.p_start(v1), p2 = prm.p_start(v2);
                    const real dox = real(gA.degree[iA]) * real(gB.degree[iB]);
                    const real dx = dox * inv1q2;
                    const real v = prm.node_kernel(v1, v2);
                    const real YDq = X[i], Yp = X2[i];
                    auto dp1 = prm.p_start._j_a_c_o_b_i_a_n_(v1);
                    auto dp2 = prm.p_start._j_a_c_o_b_i_a_n_(v2);
                    auto dv = prm.node_kernel._j_a_c_o_b_i_a_n_(v1, v2);
#pragma unroll
                    for (int j = 0; j < PStart::jac_dims; ++j)
                        jac[j] += (real(dp1[j]) * p2 + p1 * real(dp2[j])) * YDq;
                    jac[off_q] += real(2) * Q * p1 * p2 * YDq - real(2) * Q3 * Yp * dox / v * YDq;
#pragma unroll
                    for (int j = 0; j < NodeK::jac_dims; ++j)
                        jac[off_v + j] += dx * Yp * YDq / (v * v) * real(dv[j]);
                }
                if constexpr (EdgeK::jac_dims > 0) {
                    // d/d(edge theta): sum over the terms of Yp[row] YDq[column]
                    // dE_j -- the walk of the mat-vec with the rows of the first
                    // solution staged
                    for (int base = rlo; base < rhi; base += G) {
                        const int iA = base + g;
                        const bool row_ok = g < G && iA < rhi;
                        const int rsA = row_ok ? (int)gA.rowptr[iA] : 0;
                        const int dA = row_ok ? (int)gA.rowptr[iA + 1] - rsA : 0;
                        int dmax = 0;
                        for (int k = 0; k < G && base + k < rhi; ++k) {
                            const int d = (int)gA.rowptr[base + k + 1] - (int)gA.rowptr[base + k];
                            dmax = d > dmax ? d : dmax;
                        }
                        const int b0 = seg_b0;
                        const int b1 = row_ok ? seg_b1 : b0;
                        real acc[EdgeK::jac_dims];
#pragma unroll
                        for (int j = 0; j < EdgeK::jac_dims; ++j) acc[j] = 0;
                        for (int a0 = 0; a0 < dmax; a0 += A_ROWS) {
                            __syncthreads();
                            const int nv = __builtin_amdgcn_readfirstlane(dA - a0 < A_ROWS ? dA - a0 : A_ROWS);
                            edge_t eA[A_ROWS];
#pragma unroll
                            for (int u = 0; u < A_ROWS; ++u) {
                                if (u < nv) {
                                    eA[u] = gA.edge[rsA + a0 + u];
                                    const unsigned jA = gA.nz[rsA + a0 + u].j;
                                    if (lane_ok) st[lb * A_ROWS + u] = X[(size_t)jA * nB + lb];
                                }
                            }
                            __syncthreads();
                            // (full groups of four rows without tests, as in the mat-vec)
                            auto pass = [&](auto full_groups) {
                                constexpr int NF = 4 * decltype(full_groups)::value;
                                for (int b = b0; b < b1; ++b) {
                                    const edge_t eB = gB.edge[b];
                                    const unsigned col = gB.nz[b].j;
                                    real pv[A_ROWS];
                                    load_staged(st + (size_t)col * A_ROWS, pv);
#pragma unroll
                                    for (int u = 0; u < A_ROWS && u < NF + 4; ++u) {
                                        if (u < NF || u < nv) {
                                            auto de = prm.edge_kernel._j_a_c_o_b_i_a_n_(pick(sw, eA[u], eB), pick(sw, eB, eA[u]));
                                            const real w = pv[u];
#pragma unroll
                                            for (int j = 0; j < EdgeK::jac_dims; ++j) acc[j] += w * real(de[j]);
                                        }
                                    }
                                }
                            };
                            if (nv >= 8 && A_ROWS == 8) pass(std::integral_constant<int, A_ROWS == 8 ? 2 : 1>{});
                            else if (nv >= 4) pass(std::integral_constant<int, 1>{});
                            else pass(std::integral_constant<int, 0>{});
                        }
                        if (seg_ok && row_ok) {
                            const real Yp = X2[(size_t)iA * nB + iBs];
#pragma unroll
                            for (int j = 0; j < EdgeK::jac_dims; ++j) jac[off_e + j] += Yp * acc[j];
                        }
                    }
                }
                const size_t plane = (size_t)prm.nX * prm.nY;
#pragma unroll
                for (int j = 0; j < n_jac; ++j) {
                    real v[1] = {block_reduce<real, W>::sum(jac[j], red)};
                    grp.sum(v, red);
                    if (tid == 0 && grp.part == 0) {
                        if (flags & F_PACKED) {
                            prm.gradient[(size_t)prm.order[t] * n_jac + j] = v[0];
                        } else if (flags & F_DIAGONAL) {
                            prm.gradient[(size_t)I1 + (size_t)prm.nX * j] = v[0];
                        } else {
                            prm.gradient[(size_t)I1 + (size_t)prm.nX * I2 + plane * j] = v[0];
                            if (mirror) prm.gradient[(size_t)I2 + (size_t)prm.nX * I1 + plane * j] = v[0];
                        }
                    }
                }
            }
            };      // solve_pair
            if (fits) {
                if (sw_) solve_pair(std::true_type{}, std::true_type{});
                else solve_pair(std::true_type{}, std::false_type{});
            } else {
                if (sw_) solve_pair(std::false_type{}, std::true_type{});
                else solve_pair(std::false_type{}, std::false_type{});
            }
        }
    }
};

}  // namespace mgk
}  // namespace graphdot
#endif
