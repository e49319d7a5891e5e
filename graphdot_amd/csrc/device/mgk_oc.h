// Owner-computes pair solver for MI355X (gfx950, wave64): the product-graph
// mat-vec of the marginalized graph kernel in ONE stage.
//
// Same system, iteration and stopping rules as pair_solver (mgk_solver.h;
// reference: graphdot/cpp/marginalized_kernel.h:189-997,
// graphdot/kernel/marginalized/template.cu:57-474), different work split.
// pair_solver cuts the Kronecker sum of a row (i1, i2),
//     (Wp)[i1,i2] = sum_{a in adj(i1)} sum_{b in adj(i2)} E[a,b] p[j1(a), j2(b)],
// into tasks (a, i2) of deg2(i2) terms so that the 64 tasks of a wave
// instruction have one trip count, and pays for it with a second stage: the
// partial sums U[a, i2] go through LDS and every row re-reads and masks its
// deg1(i1) entries (8 VALU + 4 LDS reads per row and iteration, a third of
// the iteration's instructions, and the bank conflicts of those reads).
//
// Here the lane that owns row (i1, i2) computes the whole double sum,
// deg1(i1) * deg2(i2) terms, and the trip counts are made uniform by the
// ORDER of the rows instead: rows are dealt to lanes sorted by descending
// deg1(i1) * deg2(i2), so the 64 rows of a wave instruction share (nearly)
// one product and the wave-uniform maximum is that of its first row.  Nodes
// are stored by descending degree (packer), so the rows of one degree pair
// (d1, d2) form a rectangle of the (i1, i2) grid; the order of the
// (DMAX + 1)^2 rectangles by product is a compile-time table, their sizes
// come from two degree histograms (ballots), and the sorted position of a row
// is  off[d1][d2] + (i1 - start1[d1]) * cnt2[d2] + (i2 - start2[d2]).
//
// Per CG iteration a lane then does: one LDS gather + one FMA per nonzero
// slot, one lane-contiguous LDS store of the finished row sum per row batch
// (the batch a slot belongs to is only known at run time, the row registers
// are indexed at compile time: the sums take a lane-private round trip
// through LDS, M0-relative stores/loads without address registers), the
// fused vector updates, two wave reductions and one scattered store of p per
// row.  No second stage, no U, no masks, one barrier less.  Padding is
// higher than pair_solver's (24 slots against 17 on the QM7-like set, ideal
// 15), the instruction count per iteration a third lower.
//
// Graphs whose largest degree exceeds DMAX take pair_solver.
//
// Microkernel tables (TAB): when the labels of a call fall into few classes
// (GraphArena.classes: 19 atom x 3 bond classes on the QM7-like set) one tiny
// launch per evaluation (`fill_tables`) evaluates the node and edge
// microkernels -- and their Jacobians for the gradient solver -- once per
// pair of classes into a global table (a few KB, cache resident), and the
// pair solvers look the values up by the u8 class ids that are staged with
// the graph images: nnz1 * nnz2 + n1 * n2 table reads per pair instead of as
// many evaluations of the generated code.  (The reference evaluates the edge
// kernel once per nonzero per CG iteration, marginalized_kernel.h:299-300,346.)
#ifndef GRAPHDOT_HIP_MGK_OC_H_
#define GRAPHDOT_HIP_MGK_OC_H_
#include "mgk_solver.h"

// -DGD_MARKS: phase names as comments in the ISA (scripts/isa_phases.py counts
// the instructions between them)
#ifdef GD_MARKS
#define GD_MARK(name) asm volatile("; GDMARK " #name)
#else
#define GD_MARK(name) ((void)0)
#endif

namespace graphdot {
namespace mgk {

// (d1, d2) for d1, d2 in [0, DMAX], sorted by descending d1 * d2 (ties in
// row-major order): the order in which the degree-pair rectangles are laid
// out in the sorted row space.
template<int DMAX> struct class_order {
    constexpr static int NC = DMAX + 1, NCP = NC * NC;
    unsigned char d1[NCP], d2[NCP];
    constexpr class_order() : d1{}, d2{} {
        int n = 0;
        for (int p = DMAX * DMAX; p >= 0; --p)
            for (int a = 0; a <= DMAX; ++a)
                for (int b = 0; b <= DMAX; ++b)
                    if (a * b == p) {
                        d1[n] = (unsigned char)a;
                        d2[n] = (unsigned char)b;
                        ++n;
                    }
    }
};

// Table layout (reals): [kv: nv^2][ke: ne^2], and for C == 2 behind them
// [dkv_j: nv^2 for each node hyperparameter j][dke_j: ne^2 for each edge one].
// The representatives of the edge classes carry weight 1: tables hold the
// label part, weights are applied per nonzero.
template<class real, int C, class Graph, class NodeK, class EdgeK, class PStart>
__device__ __forceinline__ void fill_tables(params_t<real, Graph, NodeK, EdgeK, PStart> const &prm) {
    using node_t = typename Graph::node_t;
    using edge_t = typename Graph::edge_t;
    node_t const *const vr = reinterpret_cast<node_t const *>(prm.arena + prm.vrep);
    edge_t const *const er = reinterpret_cast<edge_t const *>(prm.arena + prm.erep);
    const unsigned nv = prm.n_vclass, ne = prm.n_eclass;
    const unsigned nvv = nv * nv, nee = ne * ne;
    real *const kv = prm.tables, *const ke = kv + nvv;
    real *const dkv = ke + nee, *const dke = dkv + (size_t)NodeK::jac_dims * nvv;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < nvv + nee; k += gridDim.x * blockDim.x) {
        if (k < nvv) {
            const unsigned c1 = k / nv, c2 = k - c1 * nv;
            kv[k] = real(prm.node_kernel(vr[c1], vr[c2]));
            if constexpr (C == 2) {
                auto d = prm.node_kernel._j_a_c_o_b_i_a_n_(vr[c1], vr[c2]);
#pragma unroll
                for (int j = 0; j < NodeK::jac_dims; ++j) dkv[(size_t)j * nvv + k] = real(d[j]);
            }
        } else {
            const unsigned e = k - nvv, c1 = e / ne, c2 = e - c1 * ne;
            ke[e] = real(prm.edge_kernel(er[c1], er[c2]));
            if constexpr (C == 2) {
                auto d = prm.edge_kernel._j_a_c_o_b_i_a_n_(er[c1], er[c2]);
#pragma unroll
                for (int j = 0; j < EdgeK::jac_dims; ++j) dke[(size_t)j * nee + e] = real(d[j]);
            }
        }
    }
}

// Row-batch layouts.  A lane's S register slots are consumed by its R row
// batches in order; where a batch ends is either known at run time only
// (`dynamic_layout`: a wave-uniform flush mask tested after every slot, the
// finished sum parked in a lane-private LDS cell because the row registers
// are indexed at compile time) or fixed at compile time (`seg_layout<L...>`:
// batch k owns exactly L[k] slots, so the slot -> row-register binding is
// static: no mask, no test, no branch, no LDS round trip of the row sums --
// per CG iteration 2 scalar instructions per slot and 2 LDS operations + ~3
// vector instructions per row less).  The host assigns a pair to a static
// layout when the degree product of the first row of every batch k is at most
// L[k] (HIPBackend.oc_trips); the degree products of molecular graphs fall
// into a handful of such profiles: (16, 4, 4, 1) ... on the QM7-like set.
struct dynamic_layout {
    constexpr static bool is_static = false;
};
template<int... L> struct seg_layout {
    constexpr static bool is_static = true;
    constexpr static int R = sizeof...(L);
    constexpr static int S = (L + ... + 0);
    struct tables_t {
        int end[R > 0 ? R : 1];           // one past the last slot of batch k
        bool last[S > 0 ? S : 1];         // slot s closes its batch
        constexpr tables_t() : end{}, last{} {
            constexpr int len[R > 0 ? R : 1] = {L...};
            int e = 0;
            for (int k = 0; k < R; ++k) {
                e += len[k];
                end[k] = e;
                if (len[k] > 0) last[e - 1] = true;
            }
        }
    };
    constexpr static tables_t T{};
};

// Kernel parameters of the nodal finite-difference gradient solver (NGRAD):
// the common block plus the perturbed hyperparameter sets
// state(theta_j e^{+eps}), state(theta_j e^{-eps}) of every node / edge
// hyperparameter, their unperturbed values and q e^{+-eps} -- the reference's
// node_kernel_diff_grid / edge_kernel_diff_grid / *_flat_theta __constant__
// symbols (template.cu:17-27, _backend_cuda.py:230-245,318-340) as kernel
// arguments.
template<class real, class Graph, class NodeK, class EdgeK, class PStart>
struct params_fd_t {
    params_t<real, Graph, NodeK, EdgeK, PStart> base;
    real q_plus, q_minus;
    real node_theta[NodeK::jac_dims > 0 ? NodeK::jac_dims : 1];
    real edge_theta[EdgeK::jac_dims > 0 ? EdgeK::jac_dims : 1];
    NodeK node_diff[NodeK::jac_dims > 0 ? 2 * NodeK::jac_dims : 1];
    EdgeK edge_diff[EdgeK::jac_dims > 0 ? 2 * EdgeK::jac_dims : 1];
};

// NGRAD: nodal outputs with their Jacobian as the reference defines it
// (template.cu:226-418): d/dp analytic, d/dq, d/d(node theta), d/d(edge theta)
// by central differences of re-solves at exp(log(theta) +- eps), every
// re-solve warm-started from the unperturbed solution and stopped at
// sqrt(rTr) < gtol N -- here in the same launch, right after the solve of the
// pair, instead of 2 (n_theta + 1) further launches.
// MAXIMIN: instead of the nodal matrix the launch writes the maximin graph
// distance of the pair (reference: graphdot/metric/maximin/_backend.cu:40-407),
//   d(i1,i2) = sqrt(max(0, 0.9999995 - k12 / sqrt(k1 k2))),
//   D = max(max_i1 min_i2 d, max_i2 min_i1 d),
// its hotspot (largest flat index i1 n2 + i2 among the entries equal to D) and,
// with NGRAD, the gradient -0.5 d(k12 / sqrt(k1 k2))/dtheta / (D + 1e-4) at the
// hotspot -- the min / max reductions are LDS atomics on the bit patterns of
// the non-negative float distances (order independent: deterministic), the
// n1 x n2 nodal block never leaves the CU.  k1, k2 (and their Jacobians) are
// the nodal self-similarities of the two graphs from a `diag` launch.
// Workgroup sums with ONE barrier each: the per-wave partials go to one of two
// scratch halves, and consecutive reductions alternate between them -- the
// barrier of reduction k + 1 (other half) is behind every wave's reads of
// reduction k, so half k % 2 can be rewritten at reduction k + 2 without the
// "previous readers are done" barrier block_reduce needs (5 -> 3 barriers per
// CG iteration for the multi-wave pairs).
template<class real, int W> struct alternating_reduce {
    __device__ static __forceinline__ void sum2(real &a, real &b, real *half) {
        wave::sum2(a, b);
        if constexpr (W > 1) {
            const int w = threadIdx.x / 64;
            if (wave::laneid() == 0) {
                half[2 * w] = a;
                half[2 * w + 1] = b;
            }
            __syncthreads();
            real sa = 0, sb = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                sa += half[2 * k];
                sb += half[2 * k + 1];
            }
            a = sa;
            b = sb;
        }
    }
    __device__ static __forceinline__ real sum(real a, real *half) {
        a = wave::sum(a);
        if constexpr (W > 1) {
            const int w = threadIdx.x / 64;
            if (wave::laneid() == 0) half[2 * w] = a;
            __syncthreads();
            real sa = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) sa += half[2 * k];
            a = sa;
        }
        return a;
    }
};

// EdgeK::packed_edge_t (printed by the backend when the edge record and the
// expression qualify, _backend_hip.py `declstruct2`): the record type of two
// edges side by side, for two evaluations of the microkernel per call
template<class K, class = void> struct packed_edge {
    constexpr static bool value = false;
    using type = void;
};
template<class K> struct packed_edge<K, std::void_t<typename K::packed_edge_t>> {
    constexpr static bool value = true;
    using type = typename K::packed_edge_t;
};

template<class real, int S, int R, int W, int C, bool NODAL, int DMAX, bool TAB, bool NGRAD, bool MAXIMIN, class LAY, class Graph, class NodeK, class EdgeK, class PStart>
struct oc_solver {
    constexpr static bool STATIC = LAY::is_static;
    // (Round 6, measured and dropped: static layouts of SEVERAL waves per pair
    // for graphs of degree 5-8 -- one compile-time layout per workgroup that
    // dominates the trip profile of each of its waves.  Configuration 2's
    // dynamic multi-wave launches need only 20-30 such profiles each
    // (scripts/config2_trip_profiles.py), but the static kernels are slower:
    // 84-146 ns per pair against 67 for the two-batch 8-wave launches, 123
    // against 115 for the three-batch one at two waves per SIMD, 316 with
    // packed addresses at four (272 bytes of scratch against the dynamic
    // kernel's 112) -- without the branches between the slots the scheduler
    // keeps more gathers and addresses live than the 128 registers of a
    // multi-wave workgroup hold; step 3.54 against 2.42 ms.
    // profiles/r06_static_multiwave_f32.log, commit fc7ac01.)
    static_assert(!STATIC || W == 1, "static row-batch layouts are one-wave layouts");
    // FLY (S = 0): no register slots.  The product-graph operator is NOT
    // materialised; the owner of a row walks adj(i1) x adj(i2) in every CG
    // iteration and evaluates the edge microkernel per term, as the reference
    // does (marginalized_kernel.h:299-300,346) -- for graphs whose rows have
    // hundreds of terms (from_ase-like molecular graphs: 88 % dense adjacency,
    // degree up to n - 1; up to 2.5e5 terms per pair against the 6.5e4
    // register slots of a 1024-lane workgroup).  Rows stay in natural order
    // (dense graphs have near-equal degrees: nothing to sort), any degree.
    constexpr static bool FLY = S == 0;
    static_assert(!FLY || (!STATIC && (C == 1 || !NODAL)),
                  "the on-the-fly solver: values (graph-level, nodal with their finite-difference Jacobian, maximin) and graph-level value + gradient");
    constexpr static int SA = S > 0 ? S : 1;    // slot array extent
    constexpr static int FLY_U = 4;             // terms per trip of the inner loop (2 / 4 / 8 measured: 5.06 / 4.95 / 4.33 M pairs/s before the dense product, which walks blocks of four)
    static_assert(!MAXIMIN || (NODAL && C == 1), "the maximin epilogue works on the nodal solution of a value solve");
    using P = params_t<real, Graph, NodeK, EdgeK, PStart>;
    using PF = std::conditional_t<NGRAD, params_fd_t<real, Graph, NodeK, EdgeK, PStart>, P>;
    __device__ static __forceinline__ P const &common(P const &p) { return p; }
    __device__ static __forceinline__ P const &common(params_fd_t<real, Graph, NodeK, EdgeK, PStart> const &p) { return p.base; }
    static_assert(!NGRAD || (C == 1 && NODAL && !TAB), "the nodal gradient solver is a value solver with nodal output and direct microkernel evaluation");
    using node_t = typename Graph::node_t;
    using edge_t = typename Graph::edge_t;
    constexpr static int T = 64 * W;            // threads per pair (= per workgroup)
    // DENSE (on-the-fly variants, weighted graphs, direct microkernel
    // evaluation): from_ase-like molecular graphs are 88 % dense, and a pair
    // whose adjacency matrices are more than ~60 % full is cheaper as a DENSE
    // product.  Both graphs' edge records are scattered once per pair into
    // dense n x n arrays in LDS -- weight 0 where there is no edge, graph 2's
    // TRANSPOSED -- and the owner of row (i1, i2) runs j1 over 0..n1, j2 over
    // 0..n2: no row pointers, no column indices, every lane of the workgroup
    // reads the SAME element p[j1, j2] at the same time (one LDS broadcast
    // instead of a random gather), consecutive lanes (consecutive i2) read
    // consecutive records E2T[j2][i2] (the CSR walk had the lanes of a wave
    // read the records of THEIR rows at one offset: 59 % of the LDS cycles
    // were bank conflicts), and graph 1's record is loaded once per j1.
    // ~9 issue slots per term against ~23, for (n1 n2) / (d1 d2) ~ 1.3 times
    // the terms.  Decided per pair (wave-uniform) from the four counts in the
    // graph headers.
    // Measured on the dense molecular set (profiles/sessions.md r4_session10,
    // r4_session12): float values 4.98 -> 6.31 M pairs/s, float value +
    // gradient 3.55 -> 3.93 M; 8 vector instructions per term against 19 for
    // 1.3 x the terms, bank conflicts 59 % -> 2 % of the LDS cycles, and the
    // vector pipe saturated (VALU busy 101-107 %): what is left per term is
    // the exponential and its argument (sub, mul, mul, mul, v_exp_f32).  A
    // first layout -- records in blocks of four columns per lane, for
    // ds_read_b128 -- put sixteen lanes on one bank (62 % conflict cycles,
    // 3.4 M pairs/s).  In double the software exponential dominates either
    // way and the dense form has 1.3 x as many: 1.81 against 2.26 M pairs/s,
    // off.  Nodal solves (they feed finite differences, whose two sides
    // should sum in one order) keep the CSR walk.
    constexpr static unsigned EW = sizeof(edge_t) / 4u;      // words per edge record
    constexpr static unsigned DSTRIDE = 32u;                 // words per row of the dense planes
    constexpr static bool DENSE = FLY && !TAB && GD_WEIGHTED && edge_weight<edge_t>::value && sizeof(edge_t) % 4 == 0 &&
                                  sizeof(real) == 4 && !NODAL && !NGRAD && !MAXIMIN;
    // PK2 (dense product, float): the edge microkernel on the records of two
    // columns at once -- difference, square, scale, the weights' product and
    // the products with p as packed instructions (v_pk_add_f32, v_pk_mul_f32,
    // v_pk_fma_f32), 19 vector instructions per trip of four terms against 24
    constexpr static bool PK2 = DENSE && packed_edge<EdgeK>::value && sizeof(real) == 4;
    constexpr static int NR = R * T;            // row capacity
    constexpr static int NC = DMAX + 1;         // degree classes 0..DMAX
    constexpr static int NCP = NC * NC;         // degree-pair rectangles
    constexpr static int NTAB = ((NCP + 63) / 64) * 64;
    constexpr static int NM = S > 0 ? (S + 31) / 32 : 1;    // 32-bit flush-mask words
    // slots whose loads the setup keeps in flight together (the register pins
    // that keep the scheduler from hoisting all S loads sit at chunk ends)
    constexpr static int SETUP_CHUNK = 4;
    // slot setup in one pass over running element indices (15 VALU per slot
    // instead of 29 in two passes); the two-pass form keeps fewer registers
    // live and stays where the register file is the limit
    constexpr static bool ONE_PASS = sizeof(real) == 4 || W == 1;
    // gathers in flight: 8, and 16 in the static one-wave kernels that have
    // the registers for it -- double values (130.6 -> 136.9 M pairs/s) and
    // float value + gradient (89.8 -> 93.3 M); 16 costs the dynamic double
    // gradient kernels and configuration 2's multi-wave ones 10-25 % (spills)
    constexpr static int GCH = (STATIC && W == 1 && !NODAL && (sizeof(real) == 8 || C == 2)) ? 16 : 8;
    constexpr static bool ADDTID = W == 1 && C == 1 && sizeof(real) == 4 && !STATIC && !FLY;
    // Two 16-bit LDS addresses per register (PACK): the 64-slot variants of
    // the 8- and 16-wave workgroups hold S values + S addresses = 128
    // registers per lane in float, the whole budget of a 1024-thread
    // workgroup -- everything else spilled (268 B of scratch per lane,
    // profiles/r02_c2_pmc.csv).  Packed, a slot costs one extra VALU per
    // iteration (the unpack, kept inside the loop) and half an address
    // register.  Float: the one-pass value solvers with 64 slots and 8 or 16
    // waves.  Double (round 4; two-pass setup): the 16-wave value solvers,
    // capped at 128 registers -- a slot is 2 + 1/2 registers instead of 3:
    // configuration 2's (16,40,2) 1.85 -> 1.18 ms, (16,64,3) 1.02 -> 0.88;
    // the 4- and 8-wave variants have the registers and only pay the unpack
    // (+6 ... 9 %, profiles/sessions.md r4_session8).
    constexpr static bool PACK = C == 1 && !NODAL && !STATIC && !FLY &&
                                 (sizeof(real) == 4 ? (S >= 64 && W >= 8) : W >= 16);
    constexpr static int NADR = PACK ? (S + 1) / 2 : SA;
    // SL: the values of the LAST SL slots of a lane live in a lane-private
    // LDS column instead of registers (round 4; dynamic multi-wave value
    // solvers).  A workgroup of 16 waves is capped at 128 registers per lane
    // and 40 double slots are 80 of them before addresses, rows and gathers:
    // the allocator spilled what did not fit INTO the iteration, 1.4 GB of
    // scratch traffic per launch of configuration 2's (16,40,2).  An LDS slot
    // costs one conflict-free lane-contiguous read per iteration and frees
    // one register (two in double); the workgroup's 160 KB of LDS has ~100 KB
    // to spare beside p, the row sums, the row map and the images.  The table
    // is mirrored by HIPBackend.lds_slot_bytes (LDS sizing and the LDS limit
    // of the classification).
    // Configuration 2 in double (profiles/sessions.md r4_session17), 10 slots
    // (80 KB of a 1024-lane workgroup; 12 overflow the 160 KB with (16,64,3)):
    // (16,40,2) 1.19 -> 0.97 ms, (16,64,3) 0.87 -> 0.74 ms, the step 4.95 ->
    // 4.64 ms; 4 / 6 / 8 slots: 4.76 / 4.72 / 4.67 ms.
    constexpr static int lds_slot_count() {
        if (C != 1 || NODAL || NGRAD || STATIC || FLY || W == 1) return 0;
        // (only where the registers are short by construction: in the 4- and
        // 8-wave variants and in float the allocator answered LDS slots with
        // MORE scratch operations inside the loop, 4 -> 37 in the double
        // (8,64,4), 9 -> 26 in the float (4,64,5))
        if (sizeof(real) == 8 && W == 16 && (S == 40 || S == 64)) return 10;
        return 0;
    }
    constexpr static int SL = lds_slot_count();
    constexpr static int SREG = S - SL;          // slots whose values are registers
    // LEAN (static layouts, value + gradient): the solution x lives in a
    // lane-private LDS region and p only in its published copy -- the update
    // block re-reads p (once for A p, once for the x / p update) and
    // read-modify-writes x, 3 R more LDS operations per iteration (about what
    // the static layout saved on the row sums) for 4 R reals fewer in
    // registers across the gather phase: the two-right-hand-side solver is
    // what the register file limits to two waves per SIMD in double.
    // SEQ (static layouts, value + gradient): the two right-hand sides of the
    // gradient solve -- D q^2/q0^2 and p1 (x) p2 -- are solved ONE AFTER THE
    // OTHER by the one-right-hand-side iteration of the value solver, over the
    // same register slots, instead of stacked with shared alpha / beta
    // (compute_duo, marginalized_kernel.h:492-804).  Same solutions to the
    // stopping tolerance; the stacked form keeps r, p, A p, z, x of both
    // systems in registers (108-128 VGPRs in double on top of the slots: 1.9
    // waves per SIMD, VALU 55 % / LDS 55 % busy -- a latency-bound kernel),
    // the sequential form keeps those of one system and the finished solution
    // of the other.  Each system stops at sqrt(rTr) < 1e-10 * 2N / sqrt(2), so
    // that the stacked residual is under the reference's 1e-10 * 2N.
    // (Double only: the float gradient solvers keep the stacked LEAN form,
    // 52.7 against 52.8 M pairs/s with sequential solves.)
    constexpr static bool SEQ = C == 2 && STATIC && !NODAL && W == 1 && sizeof(real) == 8;
    constexpr static int CW = SEQ ? 1 : C;      // right-hand sides per CG iteration
    // (Measured and dropped, round 4: the running solution in a lane-private
    // LDS cell instead of 2 R registers lets the four-batch kernel run three
    // waves per SIMD without scratch reloads in the loop -- and is no faster
    // there, 3.08 against 2.98 ms; at two waves it loses 9 %.)
    constexpr static bool LEAN = STATIC && C == 2 && !NODAL && !SEQ;
#ifndef GD_OC_FSCAL
#define GD_OC_FSCAL 1
#endif
    // FSCAL (double builds): the SCALARS of the iteration -- pAp, rTr, rTz and
    // with them alpha and beta -- are reduced and divided in float; vectors,
    // per-lane partial sums and every update stay double.  x and r move by the
    // same alpha (converted back to double once), so r remains the residual of
    // x to double rounding whatever alpha is: a step length that is off by
    // 6e-8 relative is a marginally sub-optimal step, not an error (measured
    // on 300 pairs of the QM7-like set: the same 16.6 iterations to 1e-8 N,
    // 24.7 against 24.0 to 1e-13 N, K within 2e-13 of the direct solve).  A
    // wave sum costs 7 instructions in float (v_add_f32_dpp) against 20 in
    // double (v_add_f64 takes no DPP operand: two v_mov_dpp per step), a
    // quotient 2 against 6: the three sums and two quotients of an iteration
    // were 52 of its 134 vector instructions.
    constexpr static bool FSCAL = GD_OC_FSCAL != 0 && sizeof(real) == 8 && !NGRAD;
    using sreal = std::conditional_t<FSCAL, float, real>;
    // (Measured and dropped, round 4: iterative refinement -- the system in
    // double, solved by the float iteration in rounds with the residual
    // recomputed in double between them: correct, 18.1 float iterations in two
    // rounds, and 147 against 153 M pairs/s of the double iteration with
    // float scalars: the float remainders of the slots and the double rows
    // make it a 206-register kernel.  DESIGN.md "tried and dropped".)
    constexpr static int NSYS = C / CW;          // solves per pair
    // DLDS (the double one-wave static value solver of SIX row batches): the
    // Jacobi diagonal and its inverse -- 2 R reals, 4 R registers, read once
    // per iteration each -- live in lane-private LDS cells ([2 R][T] in the
    // [Y] region, HIPBackend.diagonals_in_lds sizes it) instead of registers.
    // The layout needs 200 registers spill-free and ran at two waves per SIMD
    // (at 168 the allocator reloads 12 values from scratch in every
    // iteration: 0.557 -> 0.849 ms); with the 24 registers out nothing is
    // reloaded at three waves.  The kernel alone gains little (0.555 -> 0.547
    // ms: the 12 conflict-free reads cost what the third wave brings, as
    // they did on the five-batch kernel, DESIGN.md "tried and dropped") -- but
    // beside the other launches of a step it overlaps better: 168.5 / 168.0
    // -> 170.4 / 169.9 M pairs/s on the headline, alternating on one box
    // (profiles/sessions.md r5_session33).
    constexpr static bool DLDS = sizeof(real) == 8 && STATIC && W == 1 && C == 1 && !NODAL &&
                                 !NGRAD && !MAXIMIN && !FLY && R == 6;
    constexpr static bool HAS_Y = (!STATIC && !FLY) || LEAN || SEQ || DLDS;   // the [Y] region exists (SEQ: the first system's solution waits there)
    constexpr static int Y_REALS = DLDS ? 2 * R * 64 * W : R * 64 * W * C;    // its size
    // GRID (static layouts whose first batch has DMAX^2 slots): slot
    // s = u DMAX + v of the first batch is the term (u-th nonzero of row i1,
    // v-th nonzero of row i2) of the batch's row, valid if u < d1 and v < d2 --
    // every row has degrees <= DMAX, so every row of the batch fits the grid.
    // The labels, columns and table offsets are read once per u and per v
    // (2 DMAX reads) instead of once per slot (DMAX^2), and a slot is an add
    // for its gather address, an add for its table offset and a select: the
    // running-index walk costs 20 VALU per slot, and the setup is a third of
    // a pair's instructions.
    template<class L> constexpr static int grid_slots() {
        if constexpr (L::is_static && !NGRAD) return L::T.end[0] == DMAX * DMAX ? DMAX * DMAX : 0;
        else return 0;
    }
    constexpr static int G0 = grid_slots<LAY>();
    static_assert(G0 == 0 || ONE_PASS, "the grid walk belongs to the one-pass slot setup");
    template<class L> constexpr static bool layout_matches() {
        if constexpr (L::is_static) return L::S == S && L::R == R;
        else return true;
    }
    static_assert(layout_matches<LAY>(), "the static layout must have S slots in R batches");
    // does slot s close its row batch?  Static layouts: a compile-time constant
    // once the slot loops are unrolled; dynamic: bit s of the flush mask
    template<class M> __device__ static __forceinline__ bool flush_at(int s, M const &m) {
        if constexpr (STATIC) return LAY::T.last[s];
        else return (m[s / 32] >> (s % 32)) & 1u;
    }
    // The graph-level value K = sum_i pp_i x_i, pp = p1 (x) p2, needs no x:
    // x = sum_k alpha_k p_k, so K = sum_k alpha_k (pp . p_k) is accumulated
    // per lane.  Saves the R solution registers (2 R in double) -- for the
    // uniform starting probability pp is one constant.  Nodal outputs and the
    // gradient need the solution itself.
    constexpr static bool KEEP_X = NODAL || C == 2;
    constexpr static int n_jac = PStart::jac_dims + 1 + NodeK::jac_dims + EdgeK::jac_dims;
    constexpr static int off_q = PStart::jac_dims;
    constexpr static int off_v = off_q + 1;
    constexpr static int off_e = off_v + NodeK::jac_dims;
    constexpr static class_order<DMAX> ORD{};

    struct lds_t {
        real red[2][2 * W];     // two halves: see alternating_reduce
        int tab_off[NTAB];      // sorted-row offset of rectangle d1 * NC + d2
        int tab_cls[64];        // [0..NC) start1, [16..) cnt2, [32..) start2
        unsigned mm_cell[4];    // MAXIMIN: distance bits, hotspot, mirrored hotspot
    };

    __device__ static __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

    // A load for the scalar unit: the address is wave-uniform and the memory
    // (job list, graph headers) is not written while the kernel runs, which
    // the constant address space states -- the compiler cannot prove it of a
    // global pointer inside a loop that also stores results, and falls back to
    // vector loads plus v_readfirstlane, with every number derived from the
    // header computed per lane.
    template<class V> __device__ static __forceinline__ V scalar_load(V const *ptr) {
        static_assert(sizeof(V) % 4 == 0, "whole dwords");
        typedef const unsigned __attribute__((address_space(4))) *const_words;
        const_words w = (const_words)(std::uintptr_t)ptr;
        unsigned buf[sizeof(V) / 4];
#pragma unroll
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) buf[k] = w[k];
        V out;
        __builtin_memcpy(&out, buf, sizeof(V));
        return out;
    }

    // alpha and beta of the CG recurrence, a / b of two wave-uniform numbers.
    // Double: v_rcp_f64 and two Newton steps, then one product -- 6
    // instructions for a result within 1.5 ulp, against the 9 to 11 of the
    // division the compiler expands (a third refinement and a final
    // correction that the 1e-8 N stopping rule cannot see).
    __device__ static __forceinline__ float cg_ratio(float a, float b) { return a / b; }
    __device__ static __forceinline__ double cg_ratio(double a, double b) {
        double y = __builtin_amdgcn_rcp(b);
        y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
        y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
        return a * y;
    }

    // Sorted position of the row that thread `tid` (lane `lane` of wave `wv`)
    // owns in batch k.  The 64-row chunks of a batch go to the waves in snake
    // order (wave w takes chunk w in even batches, chunk W - 1 - w in odd
    // ones): the rows are sorted by cost, and dealt plainly wave 0 would hold
    // the heaviest chunk of every batch -- its slot count, which every other
    // wave waits for at the barriers, is 36 % (8 waves) / 49 % (16 waves)
    // above the mean on configuration 2, 25 % / 23 % in snake order.
    __device__ static __forceinline__ int row_pos(int k, int wv, int lane) {
        return k * T + 64 * ((k & 1) ? W - 1 - wv : wv) + lane;
    }

    struct row_t {        // the row a lane is filling slots for
        int i1, i2, rs1, rs2, d1, d2, prod;
    };
    constexpr static int GD_ = G0 ? DMAX : 1;
    struct grid_t {       // GRID: the first batch's row as 2 DMAX half-terms
        unsigned a[GD_], b[GD_];      // element indices (clamped into the row)
        unsigned j1[GD_], j2[GD_];    // lp byte address = j1[u] + j2[v]
        unsigned t1[GD_], t2[GD_];    // edge table index = t1[u] + t2[v]
        bool u[GD_], v[GD_];          // u < d1, v < d2
        int row;                      // i1 * ldp + i2
    };

    __device__ static __forceinline__ void run(PF const &full, lds_t &lds, real *dyn) {
        P const &prm = common(full);
        const int lane = wave::laneid();
        const int tid = (W == 1) ? lane : (int)threadIdx.x;
        const int wv = (W == 1) ? 0 : uni((int)(threadIdx.x / 64));
        // dynamic LDS: [p: u_capacity * C reals][Y: NR * C reals][rowmap: NR u32][G1][G2]
        // (static layouts keep the row sums in registers: no Y region, except
        // LEAN, which keeps the solution x there)
        real *const lp = dyn;
        real *const lY = lp + (size_t)prm.u_capacity * C;
        unsigned *const rowmap = reinterpret_cast<unsigned *>(lY + (HAS_Y ? (size_t)Y_REALS : (size_t)0));
        char *const lG1 = reinterpret_cast<char *>(rowmap + NR);
        char *const lG2 = lG1 + prm.g_capacity;
        // SL: [SL][T] slot values behind the second image
        [[maybe_unused]] real *const lV = reinterpret_cast<real *>(lG2 + prm.g_capacity);
        real *const red0 = lds.red[0], *const red1 = lds.red[1];
        using reduce = alternating_reduce<real, W>;
        using sreduce = alternating_reduce<sreal, W>;
        sreal *const sred0 = reinterpret_cast<sreal *>(red0), *const sred1 = reinterpret_cast<sreal *>(red1);
        const unsigned lY_off = uni((int)lds_offset(lY));
        graph_header_t const *const headers = reinterpret_cast<graph_header_t const *>(prm.arena);
        const int dump = (int)prm.u_capacity - 1;   // cell that dead rows publish to

        if constexpr (DENSE) {
            // the dense product reads cells of p that no row publishes (the
            // padding column of an even n2, up to three cells behind the
            // vector) against records of weight 0: they must be numbers
            if (prm.flags & F_DENSE) {
                for (unsigned k = tid; k < prm.u_capacity * (unsigned)C; k += T) lp[k] = real(0);
                job_sync<W>();
            }
        }
        for (unsigned t = blockIdx.x; t < prm.n_launch_jobs; t += gridDim.x) {
            const job_t job = scalar_load(prm.jobs + t);
            const graph_header_t h1 = scalar_load(headers + job.i), h2 = scalar_load(headers + job.j);
            const int n1 = h1.n_node, n2 = h2.n_node, N = n1 * n2;
            const int ldp = n2 | 1;            // odd row stride of p: banks spread
            const real q = prm.q, q0 = prm.q0;
            const real inv1q2 = real(1) / ((real(1) - q) * (real(1) - q));
            const real bscale = q * q / (q0 * q0);
            // label-class section in front of each image (_devicegraph.class_bytes):
            // u8 node classes [pad4(n)], u8 edge classes [pad4(nnz)], 16-aligned
            const unsigned nzpad1 = ((unsigned)n1 + 3u) & ~3u, nzpad2 = ((unsigned)n2 + 3u) & ~3u;
            const unsigned cb1 = TAB ? (nzpad1 + (((unsigned)h1.n_nz + 3u) & ~3u) + 15u) & ~15u : 0u;
            const unsigned cb2 = TAB ? (nzpad2 + (((unsigned)h2.n_nz + 3u) & ~3u) + 15u) & ~15u : 0u;

            GD_MARK(stage);
            // ---- stage both graph images in LDS: one global round trip --------
            job_sync<W>();  // previous pair is done with the LDS regions
            {
                typedef unsigned v4 __attribute__((ext_vector_type(4)));
                const unsigned w1 = (h1.perm + 2u * n1 - h1.degree + cb1 + 15u) / 16u;
                const unsigned w2 = (h2.perm + 2u * n2 - h2.degree + cb2 + 15u) / 16u;
                const v4 *const s1 = reinterpret_cast<const v4 *>(prm.arena + h1.degree - cb1);
                const v4 *const s2 = reinterpret_cast<const v4 *>(prm.arena + h2.degree - cb2);
                v4 *const d1 = reinterpret_cast<v4 *>(lG1);
                v4 *const d2 = reinterpret_cast<v4 *>(lG2);
                constexpr int K = 2;
                const unsigned wmax = w1 > w2 ? w1 : w2;
#pragma nounroll
                for (unsigned base = 0; base < wmax; base += K * T) {
                    v4 v1[K], v2[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const unsigned w = base + k * T + tid;
                        v1[k] = w < w1 ? s1[w] : v4{0u, 0u, 0u, 0u};
                        v2[k] = w < w2 ? s2[w] : v4{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const unsigned w = base + k * T + tid;
                        if (w < w1) d1[w] = v1[k];
                        if (w < w2) d2[w] = v2[k];
                    }
                }
                // lane-private row sums: rows of batches that own no slots read 0
                if constexpr (!STATIC && !FLY) {
#pragma unroll
                    for (int k = 0; k < R; ++k)
#pragma unroll
                        for (int c = 0; c < C; ++c) lY[(k * T + tid) * C + c] = 0;
                }
            }
            const Graph g1(lG1 + cb1 - h1.degree, h1);
            const Graph g2(lG2 + cb2 - h2.degree, h2);
            std::uint8_t const *const ncls1 = reinterpret_cast<std::uint8_t const *>(lG1);
            std::uint8_t const *const ncls2 = reinterpret_cast<std::uint8_t const *>(lG2);
            std::uint8_t const *const ecls1 = ncls1 + nzpad1;
            std::uint8_t const *const ecls2 = ncls2 + nzpad2;
            const unsigned nvc = prm.n_vclass, nec = prm.n_eclass;
            real const *const kvtab = prm.tables;
            real const *const ketab = kvtab + nvc * nvc;
            real const *const dkvtab = ketab + nec * nec;
            real const *const dketab = dkvtab + (size_t)NodeK::jac_dims * nvc * nvc;
            // microkernel value of a node pair: table or direct
            auto kappa_v = [&](int i1, int i2, node_t const &v1, node_t const &v2) -> real {
                if constexpr (TAB) return at32(kvtab, __umul24((unsigned)ncls1[i1], nvc) + ncls2[i2]);
                else return real(prm.node_kernel(v1, v2));
            };
            std::uint16_t const *const lrp1 = g1.rowptr;
            std::uint16_t const *const lrp2 = g2.rowptr;
            job_sync<W>();

            // DENSE: does this pair take the dense product?  Dense costs
            // ~9 issue slots for each of (n1 n2)^2 terms, the CSR walk ~23 for
            // each of nnz1 nnz2 N / N ... i.e. per row n1 n2 against d1 d2 terms
            [[maybe_unused]] bool dense_pair = false;
            [[maybe_unused]] edge_t *dE1 = nullptr, *dE2T = nullptr;
            if constexpr (DENSE) {
                // graph 2's array as dword rows, E2T[j2 / 4][word d of the
                // record][j2 % 4][i2] with rows of DSTRIDE words (hosts sets
                // F_DENSE for graphs of at most DSTRIDE nodes): the lanes of a
                // wave -- consecutive i2 -- read consecutive words (no bank
                // conflicts: blocks of four 8-byte records per lane put 16
                // lanes on one bank and ran at 62 % conflict cycles, 3.4 M
                // pairs/s), and all the words of the four columns of a trip
                // sit within 1 KB of ONE running address (ds_read2_b32 reaches
                // 255 words: whole planes per record word took an address
                // register and an addition per plane and trip)
                const unsigned n2p = ((unsigned)n2 + 3u) & ~3u;   // rows per plane: whole trips of four, the padding rows zero
                const unsigned nn1 = (unsigned)n1 * (unsigned)n1, nn2 = EW * n2p * DSTRIDE;
                dense_pair = (prm.flags & F_DENSE) && n2 <= (int)DSTRIDE &&
                             23u * (unsigned)h1.n_nz * (unsigned)h2.n_nz > 9u * nn1 * (unsigned)n2 * (unsigned)n2;
                dE1 = reinterpret_cast<edge_t *>(lG2 + prm.g_capacity);
                dE2T = reinterpret_cast<edge_t *>(reinterpret_cast<char *>(dE1) + (nn1 * (unsigned)sizeof(edge_t) + 15u) / 16u * 16u);
                if (dense_pair) {
                    // Where a graph has no edge the arrays hold a FILLER: the
                    // labels of the graph's first edge with weight 0.  The
                    // microkernel is evaluated on every cell, and on an
                    // all-zero record a label kernel need not be a number --
                    // Normalize(DotProduct()) is 0 / sqrt(0) there, Convolution
                    // divides by the lengths of two empty lists -- and NaN
                    // times the weight 0 stays NaN.  On the filler it is as
                    // finite as on any pair of real edges (dense pairs have
                    // edges: the test above needs n_nz > 0 on both sides).
                    edge_t fill1 = at32(g1.edge, 0u), fill2 = at32(g2.edge, 0u);
                    fill1.weight = 0;
                    fill2.weight = 0;
                    unsigned fw[EW];
                    __builtin_memcpy(fw, &fill2, sizeof(edge_t));
                    unsigned *const planes = reinterpret_cast<unsigned *>(dE2T);
                    for (unsigned k = tid; k < nn1; k += T) dE1[k] = fill1;
                    for (unsigned k = tid; k < nn2; k += T) {
                        // row of the plane array = (j2 / 4) (EW 4) + d 4 + j2 % 4
                        const unsigned d = ((k / DSTRIDE) >> 2) % EW;
                        unsigned w = fw[0];
#pragma unroll
                        for (unsigned dd = 1; dd < EW; ++dd) w = d == dd ? fw[dd] : w;
                        planes[k] = w;
                    }
                    job_sync<W>();
                    for (unsigned e = tid; e < (unsigned)h1.n_nz; e += T) {
                        const nz_t z = at32(g1.nz, e);
                        dE1[(unsigned)z.i * (unsigned)n1 + z.j] = at32(g1.edge, e);
                    }
                    for (unsigned e = tid; e < (unsigned)h2.n_nz; e += T) {
                        const nz_t z = at32(g2.nz, e);
                        const edge_t rec = at32(g2.edge, e);
                        unsigned words[EW];
                        __builtin_memcpy(words, &rec, sizeof(edge_t));
#pragma unroll
                        for (unsigned d = 0; d < EW; ++d)
                            planes[(((unsigned)z.j >> 2) * (EW * 4u) + d * 4u + ((unsigned)z.j & 3u)) * DSTRIDE + z.i] = words[d];
                    }
                    job_sync<W>();
                }
            }

            GD_MARK(rectangles);
            // ---- degree histograms -> offsets of the degree-pair rectangles ---
            // (every wave computes the same wave-uniform numbers)
            if constexpr (!FLY) {
                // (the packer leaves the degree histogram of every graph in its
                // header: scalar registers, no ballots over the row pointers)
                int cnt1[NC], cnt2[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    cnt1[c] = (int)h1.hist[c];
                    cnt2[c] = (int)h2.hist[c];
                }
                // nodes are stored by descending degree: class d starts after
                // all classes of higher degree
                int vcls = 0;
                int voff[NTAB / 64];
#pragma unroll
                for (int w = 0; w < NTAB / 64; ++w) voff[w] = 0;
                fill_classes(std::make_integer_sequence<int, NC>{}, vcls, cnt1, cnt2);
                fill_offsets(std::make_integer_sequence<int, NCP>{}, voff, cnt1, cnt2);
                if (wv == 0) {
                    lds.tab_cls[lane] = vcls;
#pragma unroll
                    for (int w = 0; w < NTAB / 64; ++w) lds.tab_off[64 * w + lane] = voff[w];
                }
            }
            job_sync<W>();

            GD_MARK(rowmap);
            // ---- sorted position of every row: rowmap[position] = (i1, i2) ----
            {
                divmod_walk row(tid, T, n2);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const bool ok = k * T + tid < N;
                    const int i1 = ok ? row.hi : 0, i2 = ok ? row.lo : 0;
                    row.next();
                    if constexpr (FLY) {      // natural order
                        if (ok) rowmap[k * T + tid] = ((unsigned)i1 << 16) | (unsigned)i2;
                    } else {
                    const int d1 = (int)lrp1[i1 + 1] - (int)lrp1[i1];
                    const int d2 = (int)lrp2[i2 + 1] - (int)lrp2[i2];
                    const int pos = lds.tab_off[d1 * NC + d2] +
                                    (i1 - lds.tab_cls[d1]) * lds.tab_cls[16 + d2] +
                                    (i2 - lds.tab_cls[32 + d2]);
                    if (ok) rowmap[pos] = ((unsigned)i1 << 16) | (unsigned)i2;
                    }
                }
            }
            job_sync<W>();

            auto open_row = [&](int kb) -> row_t {
                row_t r;
                const int pos = row_pos(kb, wv, lane);
                const bool ok = pos < N;
                const unsigned rm = rowmap[ok ? pos : 0];
                r.i1 = (int)(rm >> 16);
                r.i2 = (int)(rm & 0xFFFFu);
                r.rs1 = lrp1[r.i1];
                r.rs2 = lrp2[r.i2];
                r.d1 = ok ? (int)lrp1[r.i1 + 1] - r.rs1 : 0;
                r.d2 = ok ? (int)lrp2[r.i2 + 1] - r.rs2 : 0;
                r.prod = r.d1 * r.d2;
                return r;
            };

            // A lane walks the nonzero pairs (e1, e2) of a row, e2 fastest, as two
            // running element indices; the row is over when e1 reaches its end.
            // Dead rows (no terms, or beyond N) walk element 0 and are never valid.
            struct walk_t {
                unsigned e1, end1, last1, e2, rs2, end2;
                int row;    // i1 * ldp + i2
                __device__ __forceinline__ bool valid() const { return e1 < end1; }
                __device__ __forceinline__ unsigned a() const { return e1 < last1 ? e1 : last1; }
                __device__ __forceinline__ void next() {
                    ++e2;
                    const bool wrap = e2 == end2;
                    e2 = wrap ? rs2 : e2;
                    e1 += wrap ? 1u : 0u;
                }
            };
            auto open_walk = [&](int kb) -> walk_t {
                const row_t r = open_row(kb);
                const bool live = r.prod > 0;
                walk_t w;
                w.e1 = live ? (unsigned)r.rs1 : 0u;
                w.end1 = live ? (unsigned)(r.rs1 + r.d1) : 0u;
                w.last1 = live ? (unsigned)(r.rs1 + r.d1 - 1) : 0u;
                w.rs2 = live ? (unsigned)r.rs2 : 0u;
                w.end2 = live ? (unsigned)(r.rs2 + r.d2) : 1u;
                w.e2 = w.rs2;
                w.row = r.i1 * ldp + r.i2;
                return w;
            };

            // GRID: the first batch's row as 2 DMAX half-terms
            auto open_grid = [&](unsigned base, unsigned elem) -> grid_t {
                const row_t r = open_row(0);
                const bool live = r.prod > 0;
                grid_t g;
                g.row = r.i1 * ldp + r.i2;
#pragma unroll
                for (int k = 0; k < GD_; ++k) {
                    g.u[k] = k < r.d1;
                    g.v[k] = k < r.d2;
                    // (half-terms beyond the degree read element 0 of the row,
                    // dead rows element 0 of the graph)
                    g.a[k] = live ? (unsigned)r.rs1 + (g.u[k] ? (unsigned)k : 0u) : 0u;
                    g.b[k] = live ? (unsigned)r.rs2 + (g.v[k] ? (unsigned)k : 0u) : 0u;
                    const nz_t z1 = at32(g1.nz, g.a[k]), z2 = at32(g2.nz, g.b[k]);
                    g.j1[k] = base + __umul24((unsigned)z1.j, (unsigned)ldp) * elem;
                    g.j2[k] = (unsigned)z2.j * elem;
                    if constexpr (TAB) {
                        g.t1[k] = __umul24((unsigned)ecls1[g.a[k]], nec);
                        g.t2[k] = (unsigned)ecls2[g.b[k]];
                    }
                }
                return g;
            };

            GD_MARK(slots);
            // ---- nonzero slots owned by this thread ---------------------------
            real val[SA];
            unsigned adr[NADR];   // LDS byte address of the gathered element of p (two-pass setup: first (a << 16) | b, or ~0u; PACK: two per register)
            const unsigned lp_off = lds_offset(lp);
            constexpr unsigned ELEM = CW * sizeof(real);
            unsigned fm[NM];
#pragma unroll
            for (int w = 0; w < NM; ++w) fm[w] = 0;
            int n_slots = 0;
            if constexpr (!FLY) {
                if constexpr (STATIC) {
                    // batch k is live if it has rows: its slots end at a
                    // compile-time position
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        if (k * T < N) n_slots = LAY::T.end[k];
                } else {
                // pass 0 (wave-uniform): trip count of every row batch of this
                // wave = degree product of its first row -> flush mask
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int pos = row_pos(k, wv, 0);
                    int trip = 0;
                    if (pos < N) {
                        const unsigned rm = (unsigned)uni((int)rowmap[pos]);
                        const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                        trip = uni(((int)lrp1[i1 + 1] - (int)lrp1[i1]) * ((int)lrp2[i2 + 1] - (int)lrp2[i2]));
                    }
                    if (trip > 0) {
                        n_slots += trip;
                        const int last = n_slots - 1;      // < S (host guarantees)
#pragma unroll
                        for (int w = 0; w < NM; ++w)
                            if (last / 32 == w) fm[w] |= 1u << (last % 32);
                    }
                }
                n_slots = n_slots > S ? S : n_slots;
                }
                if constexpr (ONE_PASS) {
                    // one unrolled pass over the slots (walk_t above)
                    int kb = 0;
                    walk_t cur = G0 ? walk_t{} : open_walk(0);
                    [[maybe_unused]] grid_t grid;
                    [[maybe_unused]] edge_t ge1[GD_], ge2[GD_];
                    if constexpr (G0 > 0) {
                        grid = open_grid(lp_off, ELEM);
                        if constexpr (!TAB || (GD_WEIGHTED && edge_weight<edge_t>::value)) {
#pragma unroll
                            for (int k = 0; k < GD_; ++k) {
                                ge1[k] = at32(g1.edge, grid.a[k]);
                                ge2[k] = at32(g2.edge, grid.b[k]);
                            }
                        }
                    }
    #pragma unroll
                    for (int s = 0; s < S; ++s) {
                        // (limits the scheduler's hoisting of loads -- and with it the
                        // live registers -- to SETUP_CHUNK slots)
                        if (s % SETUP_CHUNK == 0) __builtin_amdgcn_sched_barrier(0);
                        real e;
                        bool ok;
                        unsigned col;
                        if (s < G0) {
                            const int gu = s / GD_, gv = s % GD_;
                            ok = grid.u[gu] && grid.v[gv];
                            if constexpr (TAB) {
                                e = at32(ketab, grid.t1[gu] + grid.t2[gv]);
                                if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                    e *= real(edge_weight<edge_t>::get(ge1[gu])) *
                                         real(edge_weight<edge_t>::get(ge2[gv]));
                            } else {
                                e = prm.edge_kernel(ge1[gu], ge2[gv]);
                            }
                            // (slots without a term gather p[0] like the running
                            // walk's: lanes on one address share the LDS access,
                            // clamped neighbours would add bank conflicts -- 3.43
                            // against 3.24 LDS cycles per gather, scripts/lds_sim.py)
                            col = ok ? grid.j1[gu] + grid.j2[gv] : lp_off;
                        } else {
                        ok = cur.valid();
                        const unsigned a = cur.a(), b = cur.e2;
                        const nz_t z1 = at32(g1.nz, a), z2 = at32(g2.nz, b);
                        if constexpr (TAB) {
                            e = at32(ketab, __umul24((unsigned)ecls1[a], nec) + ecls2[b]);
                            if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                e *= real(edge_weight<edge_t>::get(at32(g1.edge, a))) *
                                     real(edge_weight<edge_t>::get(at32(g2.edge, b)));
                        } else {
                            const edge_t e1 = at32(g1.edge, a), e2 = at32(g2.edge, b);
                            e = prm.edge_kernel(e1, e2);
                        }
                        col = ok ? __umul24((unsigned)z1.j, (unsigned)ldp) + (unsigned)z2.j : 0u;
                        col = lp_off + col * ELEM;
                        }
                        if (SL > 0 && s >= SREG) {
                            lV[(s - SREG) * T + tid] = ok ? e : real(0);
                        } else {
                            val[s] = ok ? e : real(0);
                        }
                        if constexpr (PACK) {
                            // (byte addresses below 64 KB: host-checked)
                            if (s % 2 == 0) adr[s / 2] = col;
                            else adr[s / 2] |= col << 16;
                        } else {
                            adr[s] = col;
                        }
                        if (s % SETUP_CHUNK == SETUP_CHUNK - 1 || s == S - 1) {
#pragma unroll
                            for (int u = s - s % SETUP_CHUNK; u <= s; ++u) {
                                if (SL > 0 && u >= SREG) {
                                    if constexpr (PACK) asm volatile("" : "+v"(adr[u / 2]));
                                    else asm volatile("" : "+v"(adr[u]));
                                } else if constexpr (PACK) asm volatile("" : "+v"(val[u]), "+v"(adr[u / 2]));
                                else asm volatile("" : "+v"(val[u]), "+v"(adr[u]));
                            }
                        }
                        if (s >= G0) cur.next();
                        if (flush_at(s, fm) && s != S - 1) {   // wave-uniform: next row batch
                            ++kb;
                            cur = open_walk(kb);
                        }
                    }
                } else {
                    // two passes (double precision with several waves per pair:
                    // fewer registers live across the microkernel evaluation)
                    // pass 1 (unrolled): the nonzero pair (a, b) of every slot
                    int kb = 0, j = 0, ja = 0, jb = 0;
                    row_t cur = open_row(0);
                    // (PACK: the (a, b) pairs of pass 1 in an array of their
                    // own, consumed slot by slot while `adr` fills up)
                    [[maybe_unused]] unsigned ab_[PACK ? SA : 1];
                    unsigned (&ab)[PACK ? SA : NADR] = select_array<PACK>(ab_, adr);
    #pragma unroll
                    for (int s = 0; s < S; ++s) {
                        ab[s] = (s < n_slots && j < cur.prod)
                            ? ((unsigned)(cur.rs1 + ja) << 16) | (unsigned)(cur.rs2 + jb) : ~0u;
                        asm volatile("" : "+v"(ab[s]));
                        ++j;
                        ++jb;
                        if (jb >= cur.d2) {
                            jb = 0;
                            ++ja;
                        }
                        if ((fm[s / 32] >> (s % 32)) & 1u) {   // wave-uniform
                            ++kb;
                            j = ja = jb = 0;
                            cur = open_row(kb);
                        }
                    }
                    // pass 2: labels -> edge-kernel value and gather index
    #pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (s % SETUP_CHUNK == 0) __builtin_amdgcn_sched_barrier(0);
                        const bool ok = ab[s] != ~0u;
                        const unsigned a = ok ? (ab[s] >> 16) : 0u, b = ok ? (ab[s] & 0xFFFFu) : 0u;
                        const nz_t z1 = at32(g1.nz, a), z2 = at32(g2.nz, b);
                        real e;
                        if constexpr (TAB) {
                            e = at32(ketab, __umul24((unsigned)ecls1[a], nec) + ecls2[b]);
                            if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                e *= real(edge_weight<edge_t>::get(at32(g1.edge, a))) *
                                     real(edge_weight<edge_t>::get(at32(g2.edge, b)));
                        } else {
                            const edge_t e1 = at32(g1.edge, a), e2 = at32(g2.edge, b);
                            e = prm.edge_kernel(e1, e2);
                        }
                        unsigned col = ok ? __umul24((unsigned)z1.j, (unsigned)ldp) + (unsigned)z2.j : 0u;
                        col = lp_off + col * ELEM;
                        if (SL > 0 && s >= SREG) {
                            lV[(s - SREG) * T + tid] = ok ? e : real(0);
                            asm volatile("" : "+v"(col));
                        } else {
                            val[s] = ok ? e : real(0);
                            asm volatile("" : "+v"(val[s]), "+v"(col));
                        }
                        if constexpr (PACK) {
                            // (byte addresses below 64 KB: at most 4096 rows of 8 bytes)
                            if (s % 2 == 0) adr[s / 2] = col;
                            else adr[s / 2] |= col << 16;
                        } else {
                            adr[s] = col;
                        }
                    }
                }
            }

            GD_MARK(rows);
            // ---- rows owned by this thread (sorted order) ----------------------
            real dg[R], mi[R], r[CW][R], p[CW][R];
            real x[C][KEEP_X ? R : 1];
            [[maybe_unused]] real xq[SEQ ? R : 1];   // SEQ: the solution of the system being solved
            real pp[KEEP_X ? 1 : R];   // p1(i1) p2(i2) of the rows
            real xs = 0;               // this lane's share of sum_i pp_i x_i
            int paddr[R];
            [[maybe_unused]] unsigned rowid[FLY ? R : 1];   // FLY: (i1 << 16) | i2, ~0u for dead rows
            real rTz = 0;
            unsigned it = 0;
            [[maybe_unused]] sreal rTr_first = 0;   // SEQ: |r|^2 the first solve ended with
            [[maybe_unused]] const unsigned lv_lane0 = lds_offset(lV) + (unsigned)tid * (unsigned)sizeof(real);

            // FLY: the owner of row (i1, i2) walks adj(i1) x adj(i2) and
            // evaluates the edge microkernel `ek` per term (per-lane trip
            // counts: the EXEC mask narrows as lanes finish); ysum[c][k] = the
            // off-diagonal sum of row batch k for right-hand side c over the
            // vector published in lp
            [[maybe_unused]] auto fly_matvec = [&](auto const &ek, real (&ysum)[CW][(STATIC || FLY) ? R : 1]) {
                if constexpr (DENSE) if (dense_pair) {      // (workgroup-uniform)
                    static_assert(FLY_U == 4, "the dense product walks blocks of four columns");
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const unsigned rm = rowid[k];
                        const bool live = rm != ~0u;
                        const unsigned i1 = live ? rm >> 16 : 0u, i2 = live ? rm & 0xFFFFu : 0u;
                        edge_t const *const e1row = dE1 + i1 * (unsigned)n1;
                        unsigned const *const e2lane = reinterpret_cast<unsigned const *>(dE2T) + i2;
                        real acc[CW];
#pragma unroll
                        for (int c = 0; c < CW; ++c) acc[c] = 0;
                        // (rows are dealt in natural order: a wave whose 64 rows of
                        // this batch all lie beyond N has nothing to sum -- the CSR
                        // walk gives dead rows zero trips, the dense loops are
                        // uniform and would run in full)
                        const bool wave_live = row_pos(k, wv, 0) < N;
                        if (wave_live)
                        for (unsigned j1 = 0; j1 < (unsigned)n1; ++j1) {
                            const edge_t e1 = e1row[j1];
                            const unsigned rowp = lp_off + __umul24(j1, (unsigned)ldp) * ELEM;
                            real part[CW][4];
#pragma unroll
                            for (int c = 0; c < CW; ++c)
#pragma unroll
                                for (int u = 0; u < 4; ++u) part[c][u] = 0;
                            // graph 1's record in both halves of a packed one
                            [[maybe_unused]] std::conditional_t<PK2, typename packed_edge<EdgeK>::type, int> e1p{};
                            if constexpr (PK2) {
                                static_assert(sizeof(e1p) == 2 * sizeof(edge_t), "packed edge record: two records, leaf by leaf");
                                unsigned w1[EW], w2[2 * EW];
                                __builtin_memcpy(w1, &e1, sizeof(edge_t));
#pragma unroll
                                for (unsigned d = 0; d < EW; ++d) w2[2 * d] = w2[2 * d + 1] = w1[d];
                                __builtin_memcpy(&e1p, w2, sizeof(e1p));
                            }
                            // one trip: four columns of graph 2 against e1.
                            // word(d * 4 + u): word d of the record of column u;
                            // pat: LDS address of p[j1, first column].  (The last
                            // trip reads up to three cells behind the row of p --
                            // the next row, the padding column or the cells
                            // behind the vector, all of them numbers (zeroed at
                            // the start of the launch, the host sizes p four cells
                            // longer): their records have weight 0.  Clamped
                            // columns cost a scalar select and a register move per
                            // term.)
                            auto trip = [&](auto const &word, unsigned pat) {
                                if constexpr (PK2) {
                                    using edge2_t = typename packed_edge<EdgeK>::type;
#pragma unroll
                                    for (int h = 0; h < 2; ++h) {
                                        unsigned w2[2 * EW];
#pragma unroll
                                        for (unsigned d = 0; d < EW; ++d) {
                                            w2[2 * d] = word(d * 4u + 2u * (unsigned)h);
                                            w2[2 * d + 1] = word(d * 4u + 2u * (unsigned)h + 1u);
                                        }
                                        edge2_t e2p;
                                        __builtin_memcpy(&e2p, w2, sizeof(edge2_t));
                                        real pe[2][CW];
                                        load_elem_at<CW>(pat + 2u * (unsigned)h * ELEM, pe[0]);   // the same address in every lane
                                        load_elem_at<CW>(pat + (2u * (unsigned)h + 1u) * ELEM, pe[1]);
                                        const auto e = ek(e1p, e2p);
#pragma unroll
                                        for (int c = 0; c < CW; ++c) {
                                            part[c][2 * h] += real(lane_of(e, 0)) * pe[0][c];
                                            part[c][2 * h + 1] += real(lane_of(e, 1)) * pe[1][c];
                                        }
                                    }
                                } else {
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        unsigned words[EW];
#pragma unroll
                                        for (unsigned d = 0; d < EW; ++d) words[d] = word(d * 4u + (unsigned)u);
                                        edge_t e2;
                                        __builtin_memcpy(&e2, words, sizeof(edge_t));
                                        real pe[CW];
                                        load_elem_at<CW>(pat + (unsigned)u * ELEM, pe);
                                        const real e = real(ek(e1, e2));
#pragma unroll
                                        for (int c = 0; c < CW; ++c) part[c][u] += e * pe[c];
                                    }
                                }
                            };
                            // (j2 is uniform: the block address and the columns
                            // of p are scalar arithmetic.  This lane's column of
                            // graph 2's records is the same for every j1; kept in
                            // registers for the row batch -- 64 of them, the trips
                            // unrolled to the 32-node limit behind uniform
                            // branches -- the kernels fell from five waves per
                            // SIMD to three and lost: 7.9 against 9.8 M pairs/s,
                            // profiles/sessions.md r4_session25.  The loop is bound
                            // by the vector pipe -- four v_exp_f32 at quarter rate
                            // are half of a trip -- not by these reads.)
                            for (unsigned j2 = 0; j2 < (unsigned)n2; j2 += 4) {
                                unsigned const *const blk = e2lane + j2 * (EW * DSTRIDE);
                                trip([&](unsigned w_) { return blk[w_ * DSTRIDE]; }, rowp + j2 * ELEM);
                            }
#pragma unroll
                            for (int c = 0; c < CW; ++c) acc[c] += (part[c][0] + part[c][1]) + (part[c][2] + part[c][3]);
                        }
#pragma unroll
                        for (int c = 0; c < CW; ++c) ysum[c][k] = live ? acc[c] : real(0);
                    }
                    return;
                }
                if constexpr (FLY) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const unsigned rm = rowid[k];
                    const bool live = rm != ~0u;
                    const unsigned i1 = live ? rm >> 16 : 0u, i2 = live ? rm & 0xFFFFu : 0u;
                    const unsigned a0 = lrp1[i1], a1 = live ? (unsigned)lrp1[i1 + 1] : a0;
                    const unsigned b0 = lrp2[i2], b1 = live ? (unsigned)lrp2[i2 + 1] : b0;
                    real acc[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = 0;
                    for (unsigned a = a0; a < a1; ++a) {
                        const edge_t e1 = at32(g1.edge, a);
                        const unsigned rowp = lp_off + __umul24((unsigned)at32(g1.nz, a).j, (unsigned)ldp) * ELEM;
                        [[maybe_unused]] unsigned c1 = 0;
                        [[maybe_unused]] real w1 = 1;
                        if constexpr (TAB) {
                            c1 = __umul24((unsigned)ecls1[a], nec);
                            if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                w1 = real(edge_weight<edge_t>::get(e1));
                        }
                        // FLY_U terms per trip, indices clamped to the row's last
                        // element and the surplus zeroed: the loads of a trip are
                        // independent (one LDS latency per trip, not per term)
                        real part[C][FLY_U];
#pragma unroll
                        for (int c = 0; c < C; ++c)
#pragma unroll
                            for (int u = 0; u < FLY_U; ++u) part[c][u] = 0;
                        const unsigned blast = b1 - 1u;     // (b1 > b0 inside the loop)
                        // (as written: with a second instantiation of the
                        // microkernel in the function -- PK2 -- the loop
                        // vectoriser interleaved two trips of this body, 168
                        // registers and 316 bytes of scratch for the whole
                        // kernel instead of 85 and none)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                        for (unsigned b = b0; b < b1; b += FLY_U) {
                            real e[FLY_U], pv[C][FLY_U];
#pragma unroll
                            for (int u = 0; u < FLY_U; ++u) {
                                const unsigned bb = b + u < b1 ? b + u : blast;
                                const unsigned col = (unsigned)at32(g2.nz, bb).j;
                                if constexpr (TAB) {
                                    e[u] = at32(ketab, c1 + ecls2[bb]);
                                    if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                        e[u] *= real(edge_weight<edge_t>::get(at32(g2.edge, bb)));
                                } else {
                                    e[u] = real(ek(e1, at32(g2.edge, bb)));
                                }
                                real pe[C];
                                load_elem_at<C>(rowp + col * ELEM, pe);
#pragma unroll
                                for (int c = 0; c < C; ++c) pv[c][u] = pe[c];
                            }
#pragma unroll
                            for (int u = 0; u < FLY_U; ++u) {
                                const real eu = (b + u < b1) ? e[u] : real(0);
#pragma unroll
                                for (int c = 0; c < C; ++c) part[c][u] += eu * pv[c][u];
                            }
                        }
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            real psum = 0;
#pragma unroll
                            for (int u = 0; u < FLY_U; ++u) psum += part[c][u];
                            acc[c] += TAB ? psum * w1 : psum;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) ysum[c][k] = acc[c];
                }
                }
            };

            auto publish = [&](real const (&v)[CW][R]) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    real e[CW];
#pragma unroll
                    for (int c = 0; c < CW; ++c) e[c] = v[c][k];
                    store_elem<CW>(lp, (unsigned)paddr[k], e);
                }
            };
            rTz = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int pos = row_pos(k, wv, lane);
                const bool ok = pos < N;
                const unsigned rm = rowmap[ok ? pos : 0];
                if constexpr (FLY) rowid[k] = ok ? rm : ~0u;
                const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                const node_t v1 = at32(g1.node, (unsigned)i1), v2 = at32(g2.node, (unsigned)i2);
                const real dx = real(at32(g1.degree, (unsigned)i1)) *
                                real(at32(g2.degree, (unsigned)i2)) * inv1q2;
                const real vx = kappa_v(i1, i2, v1, v2);
                // (double: two reciprocals of 6 instructions instead of the two
                // divisions of 11 the compiler expands -- cg_ratio above)
                const real dgk = ok ? cg_ratio(dx, vx) : real(0);
                dg[k] = (real)dgk;
                mi[k] = (real)(ok ? cg_ratio(vx, dx) : real(0));
                if constexpr (DLDS) {
                    lY[k * T + tid] = dg[k];
                    lY[(R + k) * T + tid] = mi[k];
                }
                paddr[k] = ok ? (int)__umul24((unsigned)i1, (unsigned)ldp) + i2 : dump;
                const real b = ok ? dx * bscale : real(0);
                if constexpr (SEQ) xq[k] = 0;
                else if constexpr (KEEP_X) x[0][k] = 0;
                else pp[k] = real(prm.p_start(v1)) * real(prm.p_start(v2));
                if constexpr (LEAN) {
                    const real zero[C] = {};
                    store_elem<C>(lY, k * T + tid, zero);
                }
                r[0][k] = (real)b;
                p[0][k] = r[0][k] * mi[k];
                rTz += real(r[0][k] * p[0][k]);
                if constexpr (C == 2 && !SEQ) {
                    const real bx = ok ? real(prm.p_start(v1)) * real(prm.p_start(v2)) : real(0);
                    x[1][k] = 0;
                    r[1][k] = bx;
                    p[1][k] = bx * mi[k];
                    rTz += r[1][k] * p[1][k];
                }
            }

            // SEQ: the two systems one after the other (NSYS = 2); else one pass
#pragma nounroll
            for (int sys = 0; sys < NSYS; ++sys) {
                if constexpr (SEQ) {
                    if (sys == 1) {
                        // second system: right-hand side p1 (x) p2; the
                        // diagonal and the preconditioner stay
                        rTz = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const int pos = row_pos(k, wv, lane);
                            const bool ok = pos < N;
                            const unsigned rm = rowmap[ok ? pos : 0];
                            const node_t v1 = at32(g1.node, rm >> 16), v2 = at32(g2.node, rm & 0xFFFFu);
                            const real b = ok ? real(prm.p_start(v1)) * real(prm.p_start(v2)) : real(0);
                            xq[k] = 0;
                            r[0][k] = b;
                            p[0][k] = b * mi[k];
                            rTz += r[0][k] * p[0][k];
                        }
                    }
                }
                const real tol = (C == 2) ? real(1e-10) * real(2 * N) : prm.ftol * real(N);
                // SEQ: the stacked rule rTr_0 + rTr_1 < tol^2 -- half the budget
                // for the first system, what it left for the second
                sreal tol2 = (sreal)(tol * tol);
                // (FSCAL compares float sums: a tolerance whose square is under
                // float's normal range -- ftol below ~1e-19 / N -- must not
                // round to 0, or no residual ever passes and every pair runs to
                // the iteration cap)
                if constexpr (FSCAL) tol2 = tol2 < sreal(1.17549435e-38f) ? sreal(1.17549435e-38f) : tol2;
                if constexpr (SEQ) tol2 = sys == 0 ? tol2 * sreal(0.5) : tol2 - rTr_first;
                publish(p);
                job_sync<W>();   // the previous pair's last reduction is read
                sreal rTz_s = sreduce::sum((sreal)rTz, sred1);

                unsigned its = 0;
                if constexpr (SEQ) {
                    // (a definition of every slot register in front of the
                    // loop: what the allocator spills of them around the
                    // derivative pass is reloaded HERE, not use by use inside
                    // the iteration -- it had put eight serialised
                    // scratch_load + s_waitcnt vmcnt(0) pairs into the loop)
#pragma unroll
                    for (int s_ = 0; s_ < S; ++s_) asm volatile("" : "+v"(val[s_]), "+v"(adr[s_]));
                }
                GD_MARK(cg_loop);
                // (the reference stops after N iterations at the latest -- where CG
                // in exact arithmetic has the solution.  With step lengths rounded
                // to float (FSCAL) a system of one or two rows is left with the
                // rounding of its last step, 6e-8 of the value, when the N
                // iterations are used up: K of two one-node graphs came out as
                // float(4/3) (scripts/fuzz_parity.py, seeds 4-6).  The double
                // build iterates on to its tolerance: one or two more steps.)
                const unsigned max_its = FSCAL ? 2u * (unsigned)N + 16u : (unsigned)N;
                for (; its < max_its && rTz_s != sreal(0); ++its) {
                    job_sync<W>();   // p published
                    // row sums: sum over the slots of a batch, flushed to the
                    // lane-private cell Y[batch][lane] at wave-uniform positions
                    [[maybe_unused]] real ys[CW][(STATIC || FLY) ? R : 1];   // static layouts / FLY: the row sums
                    // (static layouts of up to seven batches: every batch is
                    // flushed below, or zeroed where the walk ends early --
                    // zeroing all of them up here was 4 to 7 register moves in
                    // every iteration and 36 bytes of scratch in two kernels:
                    // 151.2 -> 159.1 M pairs/s on the double headline, the
                    // two-batch kernel 0.108 -> 0.062 ms; the nine-batch kernel
                    // lost, 0.035 -> 0.075 ms, and keeps the zeroing up here --
                    // and so do the float kernels: 238.0 -> 231.0 M pairs/s,
                    // value + gradient 98.2 -> 95.8 with it.
                    // profiles/sessions.md r5_session15)
                    constexpr bool YS_LATE = STATIC && R <= 7 && sizeof(real) == 8;
                    if constexpr (FLY || (STATIC && !YS_LATE)) {
#pragma unroll
                        for (int k = 0; k < R; ++k)
#pragma unroll
                            for (int c = 0; c < CW; ++c) ys[c][k] = 0;
                    }
                    if constexpr (FLY) {
                        fly_matvec(prm.edge_kernel, ys);
                    } else {
                        real acc[CW];
#pragma unroll
                        for (int c = 0; c < CW; ++c) acc[c] = 0;
                        int kb = 0;
                        unsigned fmv[NM];
#pragma unroll
                        for (int w = 0; w < NM; ++w) {
                            // (readfirstlane: where the allocator has parked
                            // the mask in a vector register -- the nodal-
                            // gradient kernel of a rational-quadratic composite
                            // -- the scalar pin below alone is an "illegal VGPR
                            // to SGPR copy" and the JIT fails at run time)
                            fmv[w] = (unsigned)__builtin_amdgcn_readfirstlane((int)fm[w]);
                            asm volatile("" : "+s"(fmv[w]));
                        }
#pragma unroll
                        for (int s0 = 0; s0 < S; s0 += GCH) {
                            if (s0 >= n_slots) {        // wave-uniform: no slots left
                                if constexpr (YS_LATE) {
                                    // (the batches that are not flushed in front
                                    // of s0 -- a constant of the unrolled chunk --
                                    // own no rows)
#pragma unroll
                                    for (int k = 0; k < R; ++k)
                                        if (LAY::T.end[k] > s0) {
#pragma unroll
                                            for (int c = 0; c < CW; ++c) ys[c][k] = 0;
                                        }
                                }
                                break;
                            }
                            // (static layouts have no branch between the chunks:
                            // without the fence the scheduler merges their gathers
                            // -- 16 instead of 8 vectors in flight, and spills)
                            if constexpr (STATIC) __builtin_amdgcn_sched_barrier(0);
                            real g[CW][GCH];
                            [[maybe_unused]] real vl[SL > 0 ? GCH : 1];   // SL: this chunk's LDS-resident values
                            // (the lane's column address is re-defined here: a
                            // loop-invariant LDS load would be hoisted out of the
                            // iteration -- back into the registers it was to free)
                            [[maybe_unused]] unsigned lv_lane = lv_lane0;
                            if (SL > 0 && s0 + GCH > SREG) asm volatile("" : "+v"(lv_lane));
#pragma unroll
                            for (int jj = 0; jj < GCH; ++jj) {
                                if (SL > 0 && s0 + jj >= SREG && s0 + jj < S)
                                    vl[jj] = load_real_at<real>(lv_lane + (unsigned)((s0 + jj - SREG) * T * (int)sizeof(real)));
                                real e[CW];
#pragma unroll
                                for (int c = 0; c < CW; ++c) e[c] = 0;
                                if (s0 + jj < S) {
                                    if constexpr (PACK) {
                                        // (the unpack -- one VALU -- as a volatile
                                        // instruction: it stays in the loop instead
                                        // of S hoisted registers, and without the
                                        // register copy a pinned operand costs)
                                        unsigned col;
                                        if ((s0 + jj) & 1)
                                            asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(col) : "v"(adr[(s0 + jj) / 2]));
                                        else
                                            asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(col) : "v"(adr[(s0 + jj) / 2]));
                                        load_elem_at<CW>(col, e);
                                    } else {
                                        load_elem_at<CW>(adr[s0 + jj], e);
                                    }
                                }
#pragma unroll
                                for (int c = 0; c < CW; ++c) g[c][jj] = e[c];
                            }
#pragma unroll
                            for (int jj = 0; jj < GCH; ++jj) {
                                const int s = s0 + jj;
                                if (s < S) {
#pragma unroll
                                    for (int c = 0; c < CW; ++c)
                                        acc[c] += ((SL > 0 && s >= SREG) ? vl[jj] : val[s < SREG ? s : 0]) * g[c][jj];
                                    if (flush_at(s, fmv)) {   // wave-uniform
                                        if constexpr (STATIC) {
#pragma unroll
                                            for (int c = 0; c < CW; ++c) {
                                                ys[c][kb] = acc[c];
                                                acc[c] = 0;
                                            }
                                        } else if constexpr (ADDTID) {
                                            store_lane_contiguous<0>(lY_off + kb * (T * 4), (float)acc[0]);
                                            acc[0] = 0;
                                        } else {
                                            store_elem<C>(lY, kb * T + tid, acc);
#pragma unroll
                                            for (int c = 0; c < C; ++c) acc[c] = 0;
                                        }
                                        ++kb;
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (LEAN) {
                        // update block from LDS-resident p and x (one wave: its LDS
                        // operations execute in order, no barriers)
                        real pAp = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            real pk[C];
                            load_elem<C>(lp, (unsigned)paddr[k], pk);
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                ys[c][k] = dg[k] * pk[c] - ys[c][k];   // (A p)
                                pAp += pk[c] * ys[c][k];
                            }
                        }
                        pAp = reduce::sum(pAp, red0);
                        if (pAp == real(0)) break;
                        const real alpha = cg_ratio((real)rTz_s, pAp);
                        real rTr = 0, rTz_next = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k)
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                r[c][k] -= alpha * ys[c][k];
                                rTr += r[c][k] * r[c][k];
                                rTz_next += r[c][k] * (mi[k] * r[c][k]);
                            }
                        reduce::sum2(rTr, rTz_next, red1);
                        real beta = cg_ratio(rTz_next, (real)rTz_s);
                        asm volatile("" : "+v"(beta));
                        // x += alpha p (also in the iteration that ends the loop,
                        // like the register form) and p = z + beta p, in place
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            real pk[C], xv[C];
                            load_elem<C>(lp, (unsigned)paddr[k], pk);
                            load_elem<C>(lY, k * T + tid, xv);
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                xv[c] += alpha * pk[c];
                                pk[c] = mi[k] * r[c][k] + beta * pk[c];
                            }
                            store_elem<C>(lY, k * T + tid, xv);
                            store_elem<C>(lp, (unsigned)paddr[k], pk);
                        }
                        if (rTr < (real)tol2) {   // sqrt(rTr) < tol
                            ++its;
                            break;
                        }
                        rTz_s = (sreal)rTz_next;
                        continue;
                    }
                    // (no barrier: a lane reads back what it wrote itself, and the
                    // LDS operations of one wave execute in order)
                    real Ap[CW][R];
                    real pAp = 0;
                    if constexpr (DLDS) {
                        unsigned ly_lane = lY_off + (unsigned)tid * (unsigned)sizeof(real);
                        asm volatile("" : "+v"(ly_lane));
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            dg[k] = load_real_at<real>(ly_lane + (unsigned)(k * T * (int)sizeof(real)));
                            mi[k] = load_real_at<real>(ly_lane + (unsigned)((R + k) * T * (int)sizeof(real)));
                        }
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        real y[CW];
                        if constexpr (STATIC || FLY) {
#pragma unroll
                            for (int c = 0; c < CW; ++c) y[c] = ys[c][k];
                        } else {
                            load_elem<C>(lY, k * T + tid, y);
                        }
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            Ap[c][k] = dg[k] * p[c][k] - y[c];
                            pAp += p[c][k] * Ap[c][k];
                        }
                    }
                    const sreal pAp_s = sreduce::sum((sreal)pAp, sred0);
                    if (pAp_s == sreal(0)) break;
                    const real alpha = (real)cg_ratio(rTz_s, pAp_s);
                    real rTr = 0, rTz_next = 0;
                    real z[CW][R];
                    if constexpr (!KEEP_X) {
                        real pdot = 0;   // (dead rows carry p = 0)
#pragma unroll
                        for (int k = 0; k < R; ++k) pdot += pp[k] * p[0][k];
                        xs += alpha * pdot;
                        // (here, not after the update of p: the old p can then be
                        // overwritten in place instead of being copied)
                        asm volatile("" : "+v"(xs));
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k)
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            if constexpr (SEQ) xq[k] += alpha * p[c][k];
                            else if constexpr (KEEP_X) x[c][k] += alpha * p[c][k];
                            r[c][k] -= alpha * Ap[c][k];
                            z[c][k] = mi[k] * r[c][k];
                            rTr += r[c][k] * r[c][k];
                            rTz_next += r[c][k] * z[c][k];
                        }
                    if constexpr (SEQ) {
                        // (x += alpha p is done HERE: moved behind the update of
                        // p by the scheduler it keeps the old p alive, a copy
                        // per row -- as for xs above)
#pragma unroll
                        for (int k = 0; k < R; ++k) asm volatile("" : "+v"(xq[k]));
                    }
                    sreal rTr_s = (sreal)rTr, rTz_next_s = (sreal)rTz_next;
                    sreduce::sum2(rTr_s, rTz_next_s, sred1);
                    if (rTr_s < tol2) {   // sqrt(rTr) < tol
                        if constexpr (SEQ) rTr_first = rTr_s;
                        ++its;
                        break;
                    }
                    real beta = (real)cg_ratio(rTz_next_s, rTz_s);
                    // (one scalar: without the pin fast-math turns z + beta p into
                    // (p rTz') (1 / rTz) + z, a multiplication more per element)
                    asm volatile("" : "+v"(beta));
#pragma unroll
                    for (int k = 0; k < R; ++k)
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            // (double: the fused multiply-add written INTO p's
                            // registers; the compiler accumulates into z's and
                            // moves the result over, R v_mov_b64 per iteration:
                            // 151.2 -> 155.2 M pairs/s alone, 159 -> 160 on top of
                            // the late zeroing of the row sums)
                            // (not the sequential value + gradient solves: there
                            // the allocator splits p around the instruction, 2 R
                            // moves instead of none)
                            if constexpr (sizeof(real) == 8 && STATIC && W == 1)
                                asm("v_fma_f64 %0, %1, %0, %2" : "+v"(p[c][k]) : "v"(beta), "v"(z[c][k]));
                            else
                                p[c][k] = z[c][k] + beta * p[c][k];
                        }
                    // (W > 1: the barriers inside the reductions above are behind
                    // every wave's gathers of the old p)
                    publish(p);
                    rTz_s = rTz_next_s;
                }
                it += its;
                if constexpr (SEQ) {
                    // the first solution waits in a lane-private LDS cell (its
                    // R reals would be live across the whole second solve)
                    if (sys == 0) {
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            lY[k * T + tid] = xq[k];
                        }
                    }
                }
            }
            if constexpr (SEQ) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    x[1][k] = xq[k];
                    x[0][k] = lY[k * T + tid];
                }
            }
            if constexpr (SEQ) it = (it + 1u) / 2u;   // (iterations per system)
            GD_MARK(epilogue);
            if (prm.iters != nullptr && tid == 0) prm.iters[prm.order[t]] = it;
            if constexpr (LEAN) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    real xv[C];
                    load_elem<C>(lY, k * T + tid, xv);
#pragma unroll
                    for (int c = 0; c < C; ++c) x[c][k] = xv[c];
                }
            }

            // ---- output ------------------------------------------------------
            const unsigned flags = prm.flags;
            const unsigned I1 = prm.starts[job.i], I2 = prm.starts[job.j];
            const bool mirror = (flags & F_SYMMETRIC) && job.i != job.j;
            real ksum = 0;
            if constexpr (!KEEP_X) {
                ksum = xs;
                if (flags & F_LMIN1) {   // minus sum_i pp_i kappa_v(i) q^2/q0^2
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const int pos = row_pos(k, wv, lane);
                        const bool ok = pos < N;
                        const unsigned rm = rowmap[ok ? pos : 0];
                        const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                        const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                        if (ok) ksum -= kappa_v(i1, i2, v1, v2) * bscale * pp[k];
                    }
                }
            } else {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int pos = row_pos(k, wv, lane);
                const bool ok = pos < N;
                const unsigned rm = rowmap[ok ? pos : 0];
                const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                real xi = x[0][k];
                if (flags & F_LMIN1) xi -= kappa_v(i1, i2, v1, v2) * bscale;
                const real pp_ = real(prm.p_start(v1)) * real(prm.p_start(v2));
                const real rv = ok ? xi * pp_ : real(0);
                ksum += rv;
                if constexpr (NODAL && !MAXIMIN) if ((flags & F_NODAL) && ok) {
                    const unsigned o1 = g1.perm[i1], o2 = g2.perm[i2];
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
            }

            // ---- maximin distance of the pair ---------------------------------
            // (graph-level output: I1, I2 are graph indices here)
            [[maybe_unused]] int mm_own[R];          // this lane's row k is the hotspot
            [[maybe_unused]] real mm_k12 = 0, mm_rs = 0, mm_k1 = 0, mm_k2 = 0, mm_D = 0;
            // hotspot row of this lane: p1 p2, the lmin correction, and the
            // raw solution of the last perturbed solve (reference_compat)
            [[maybe_unused]] real mm_pp = 0, mm_corr = 0, mm_xlast = 0;
            [[maybe_unused]] unsigned mm_n1 = 0, mm_n2 = 0;   // node offsets of the graphs
            [[maybe_unused]] unsigned mm_hot = 0;             // flat index of the hotspot
            if constexpr (MAXIMIN) {
                // row / column minima in the region of p (dead between the solve
                // and the re-solves; n1 + n2 <= n1 (n2 | 1) + 1 <= u_capacity words
                // for every n1, n2 >= 1 -- the row-sum region holds NR words in
                // float, one short of a 1 x NR pair), the three cells static
                unsigned *const dmin1 = reinterpret_cast<unsigned *>(lp);
                unsigned *const dmin2 = dmin1 + n1;
                unsigned *const cell = lds.mm_cell;     // [0] D bits, [1] hotspot, [2] mirrored hotspot
                const unsigned NS1 = prm.node_starts[job.i], NS2 = prm.node_starts[job.j];
                mm_n1 = NS1;
                mm_n2 = NS2;
                job_sync<W>();
                for (int i = tid; i < n1 + n2; i += T) dmin1[i] = 0x7F7FFFFFu;   // FLT_MAX
                if (tid < 3) cell[tid] = 0u;
                job_sync<W>();
                float dloc[R];
                real k12v[R], ppv[R], corrv[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int pos = row_pos(k, wv, lane);
                    const bool ok = pos < N;
                    const unsigned rm = rowmap[ok ? pos : 0];
                    const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                    const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                    real xi = x[0][k];
                    if (flags & F_LMIN1) xi -= kappa_v(i1, i2, v1, v2) * bscale;
                    k12v[k] = xi * real(prm.p_start(v1)) * real(prm.p_start(v2));
                    ppv[k] = real(prm.p_start(v1)) * real(prm.p_start(v2));
                    corrv[k] = x[0][k] - xi;
                    const real k1 = prm.diag[NS1 + g1.perm[i1]], k2 = prm.diag[NS2 + g2.perm[i2]];
                    const real dd = real(0.9999995f) - k12v[k] * graphdot::rsqrt(k1 * k2);
                    dloc[k] = ok ? sqrtf((float)(dd > real(0) ? dd : real(0))) : 0.f;
                    if (ok) {
                        __hip_atomic_fetch_min(dmin1 + i1, __float_as_uint(dloc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_min(dmin2 + i2, __float_as_uint(dloc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                job_sync<W>();
                for (int i = tid; i < n1 + n2; i += T)
                    __hip_atomic_fetch_max(cell, dmin1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                job_sync<W>();
                const unsigned Dbits = cell[0];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int pos = row_pos(k, wv, lane);
                    if (pos < N && __float_as_uint(dloc[k]) == Dbits) {
                        const unsigned rm = rowmap[pos];
                        const unsigned o1 = g1.perm[rm >> 16], o2 = g2.perm[rm & 0xFFFFu];
                        // flat indices + 1: 0 means "none yet"
                        __hip_atomic_fetch_max(cell + 1, o1 * (unsigned)n2 + o2 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_max(cell + 2, o2 * (unsigned)n1 + o1 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                job_sync<W>();
                const unsigned hot = cell[1] - 1u, hot_m = cell[2] - 1u;
                mm_hot = hot;
                mm_D = real(__uint_as_float(Dbits));
                if (tid == 0) {
                    prm.gramian[(size_t)I1 + (size_t)prm.nX * I2] = mm_D;
                    prm.hotspot[(size_t)I1 + (size_t)prm.nX * I2] = (std::int32_t)hot;
                    if (mirror) {
                        prm.gramian[(size_t)I2 + (size_t)prm.nX * I1] = mm_D;
                        prm.hotspot[(size_t)I2 + (size_t)prm.nX * I1] = (std::int32_t)hot_m;
                    }
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int pos = row_pos(k, wv, lane);
                    const unsigned rm = rowmap[pos < N ? pos : 0];
                    const unsigned o1 = g1.perm[rm >> 16], o2 = g2.perm[rm & 0xFFFFu];
                    mm_own[k] = pos < N && o1 * (unsigned)n2 + o2 == hot;
                    if (mm_own[k]) {
                        mm_k12 = k12v[k];
                        mm_pp = ppv[k];
                        mm_corr = corrv[k];
                        mm_xlast = x[0][k];
                        mm_k1 = prm.diag[NS1 + o1];
                        mm_k2 = prm.diag[NS2 + o2];
                        mm_rs = graphdot::rsqrt(mm_k1 * mm_k2);
                    }
                }
                job_sync<W>();   // the region of p is published to again below
            }
            if (!NODAL || !(flags & F_NODAL)) {
                job_sync<W>();   // (the loop may have left through either half)
                ksum = reduce::sum(ksum, red0);
                if (tid == 0) {
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

            // ---- nodal Jacobian by warm-started finite differences -------------
            if constexpr (NGRAD) {
                const size_t plane = (flags & F_DIAGONAL) ? (size_t)prm.nX
                                                          : (size_t)prm.nX * prm.nY;
                // J[row of the output, column col] = v for the rows of this lane
                auto write_column = [&](int col, real const (&v)[R]) {
                    if constexpr (MAXIMIN) {
                        // v = d k12 / d theta_col per row: the hotspot's owner
                        // parks it in the output; `maximin_gradient` below turns
                        // the columns into the gradient of the distance once
                        // every perturbed solve is done (_backend.cu:380-402)
#pragma unroll
                        for (int k = 0; k < R; ++k)
                            if (mm_own[k])
                                prm.gradient[(size_t)I1 + (size_t)prm.nX * I2 + plane * col] = v[k];
                        return;
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const int pos = row_pos(k, wv, lane);
                        if (pos < N) {
                            const unsigned rm = rowmap[pos];
                            const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                            const unsigned o1 = g1.perm[i1], o2 = g2.perm[i2];
                            if (flags & F_DIAGONAL) {
                                if (o1 == o2) prm.gradient[(size_t)(I1 + o1) + plane * col] = v[k];
                            } else {
                                prm.gradient[(size_t)(I1 + o1) + (size_t)prm.nX * (I2 + o2) + plane * col] = v[k];
                                if (mirror)
                                    prm.gradient[(size_t)(I2 + o2) + (size_t)prm.nX * (I1 + o1) + plane * col] = v[k];
                            }
                        }
                    }
                };
                // start-probability columns: analytic, on the (lmin-corrected)
                // output like the reference (template.cu:258-284 after :134-142)
                real ppr[R];
                {
                    real col[PStart::jac_dims > 0 ? PStart::jac_dims : 1][R];
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const int pos = row_pos(k, wv, lane);
                        const bool ok = pos < N;
                        const unsigned rm = rowmap[ok ? pos : 0];
                        const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                        const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                        const real p1 = prm.p_start(v1), p2 = prm.p_start(v2);
                        ppr[k] = ok ? p1 * p2 : real(0);
                        real xi = x[0][k];
                        if (flags & F_LMIN1) xi -= kappa_v(i1, i2, v1, v2) * bscale;
                        auto dp1 = prm.p_start._j_a_c_o_b_i_a_n_(v1);
                        auto dp2 = prm.p_start._j_a_c_o_b_i_a_n_(v2);
#pragma unroll
                        for (int j = 0; j < PStart::jac_dims; ++j)
                            col[j][k] = xi * (p1 * real(dp2[j]) + p2 * real(dp1[j]));
                    }
#pragma unroll
                    for (int j = 0; j < PStart::jac_dims; ++j) write_column(j, col[j]);
                }
                real x0[R], bv[R];
#pragma unroll
                for (int k = 0; k < R; ++k) x0[k] = x[0][k];

                // FLY: the edge microkernel of the system being solved
                [[maybe_unused]] EdgeK const *ek_cur = &prm.edge_kernel;
                // y[k] = sum over the slots of row batch k of val * (vector in lp)
                auto matvec = [&](real (&y)[R]) {
                    if constexpr (FLY) {
                        real ysum[C][R];
                        fly_matvec(*ek_cur, ysum);
#pragma unroll
                        for (int k = 0; k < R; ++k) y[k] = ysum[0][k];
                        return;
                    }
                    real acc = 0;
                    int kb = 0;
                    if constexpr (STATIC) {
#pragma unroll
                        for (int k = 0; k < R; ++k) y[k] = 0;
                    }
#pragma unroll
                    for (int s0 = 0; s0 < S; s0 += GCH) {
                        if (s0 >= n_slots) break;
                        real g[GCH];
#pragma unroll
                        for (int jj = 0; jj < GCH; ++jj)
                            g[jj] = (s0 + jj < S) ? load_real_at<real>(adr[s0 + jj < S ? s0 + jj : 0]) : real(0);
#pragma unroll
                        for (int jj = 0; jj < GCH; ++jj) {
                            const int s = s0 + jj;
                            if (s < S) {
                                acc += val[s] * g[jj];
                                if (flush_at(s, fm)) {
                                    if constexpr (STATIC) y[kb] = acc;
                                    else lY[kb * T + tid] = acc;
                                    acc = 0;
                                    ++kb;
                                }
                            }
                        }
                    }
                    if constexpr (!STATIC) {
#pragma unroll
                        for (int k = 0; k < R; ++k) y[k] = lY[k * T + tid];
                    }
                };
                auto publish1 = [&](real const (&v)[R]) {
#pragma unroll
                    for (int k = 0; k < R; ++k) lp[paddr[k]] = v[k];
                };
                // rows of the system for node microkernel `nk` and stopping
                // probability qv: diagonal, its inverse, right-hand side
                // (q0 moves with q: b = Dx, template.cu:292-300)
                auto set_rows = [&](auto const &nk, real qv) {
                    const real s = real(1) / ((real(1) - qv) * (real(1) - qv));
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const int pos = row_pos(k, wv, lane);
                        const bool ok = pos < N;
                        const unsigned rm = rowmap[ok ? pos : 0];
                        const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                        const real dx = real(g1.degree[i1]) * real(g2.degree[i2]) * s;
                        const real vx = real(nk(g1.node[i1], g2.node[i2]));
                        dg[k] = ok ? dx / vx : real(0);
                        mi[k] = ok ? vx / dx : real(0);
                        bv[k] = ok ? dx * bscale : real(0);
                    }
                };
                // slot values for edge microkernel `ek`
                auto set_vals = [&](EdgeK const &ek) {
                    if constexpr (FLY) {      // evaluated per term: just name it
                        ek_cur = &ek;
                        return;
                    }
                    int kb = 0, j = 0, ja = 0, jb = 0;
                    row_t cur = open_row(0);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (s < n_slots) {
                            const bool ok = j < cur.prod;
                            const int a = ok ? cur.rs1 + ja : 0, b = ok ? cur.rs2 + jb : 0;
                            const real e = ek(g1.edge[a], g2.edge[b]);
                            val[s] = ok ? e : real(0);
                            ++j;
                            ++jb;
                            if (jb >= cur.d2) {
                                jb = 0;
                                ++ja;
                            }
                            if (flush_at(s, fm) && s != S - 1) {
                                ++kb;
                                j = ja = jb = 0;
                                cur = open_row(kb);
                            }
                        }
                    }
                };
                // PCG on the current (dg, mi, val, bv), warm-started from xw
                const real gt = prm.gtol * real(N), gt2 = gt * gt;
                auto pcg_warm = [&](real (&xw)[R]) {
                    real rr[R], pv[R], y[R];
                    job_sync<W>();
                    publish1(xw);
                    job_sync<W>();
                    matvec(y);
                    real rz = 0;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        rr[k] = bv[k] - (dg[k] * xw[k] - y[k]);
                        pv[k] = mi[k] * rr[k];
                        rz += rr[k] * pv[k];
                    }
                    rz = reduce::sum(rz, red1);
                    for (unsigned itw = 0; itw < (unsigned)N && rz != real(0); ++itw) {
                        job_sync<W>();
                        publish1(pv);
                        job_sync<W>();
                        matvec(y);
                        real pAp = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            y[k] = dg[k] * pv[k] - y[k];
                            pAp += pv[k] * y[k];
                        }
                        pAp = reduce::sum(pAp, red0);
                        if (pAp == real(0)) break;
                        const real alpha = rz / pAp;
                        real rTr = 0, rz_next = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            xw[k] += alpha * pv[k];
                            rr[k] -= alpha * y[k];
                            rTr += rr[k] * rr[k];
                            rz_next += rr[k] * rr[k] * mi[k];
                        }
                        reduce::sum2(rTr, rz_next, red1);
                        if (rTr < gt2) break;
                        const real beta = rz_next / rz;
#pragma unroll
                        for (int k = 0; k < R; ++k) pv[k] = mi[k] * rr[k] + beta * pv[k];
                        rz = rz_next;
                    }
                };
                // one central difference: solve(+), solve(-), write the column
                auto difference = [&](int col, real denom, auto &&perturb_plus, auto &&perturb_minus) {
                    real xa[R], xb[R];
                    perturb_plus();
#pragma unroll
                    for (int k = 0; k < R; ++k) xa[k] = x0[k];
                    pcg_warm(xa);
                    perturb_minus();
#pragma unroll
                    for (int k = 0; k < R; ++k) xb[k] = x0[k];
                    pcg_warm(xb);
                    if constexpr (MAXIMIN) {
#pragma unroll
                        for (int k = 0; k < R; ++k)
                            if (mm_own[k]) mm_xlast = xb[k];
                    }
                    const real inv = real(1) / denom;
#pragma unroll
                    for (int k = 0; k < R; ++k) xa[k] = (xa[k] - xb[k]) * inv * ppr[k];
                    write_column(col, xa);
                };
                // d/dq
                difference(off_q, real(2) * prm.eps * q,
                           [&] { set_rows(prm.node_kernel, full.q_plus); },
                           [&] { set_rows(prm.node_kernel, full.q_minus); });
                // d/d(node hyperparameters)
                for (int j = 0; j < NodeK::jac_dims; ++j)
                    difference(off_v + j, real(2) * prm.eps * full.node_theta[j],
                               [&] { set_rows(full.node_diff[2 * j], q); },
                               [&] { set_rows(full.node_diff[2 * j + 1], q); });
                // d/d(edge hyperparameters)
                if constexpr (EdgeK::jac_dims > 0) set_rows(prm.node_kernel, q);
                for (int j = 0; j < EdgeK::jac_dims; ++j)
                    difference(off_e + j, real(2) * prm.eps * full.edge_theta[j],
                               [&] { set_vals(full.edge_diff[2 * j]); },
                               [&] { set_vals(full.edge_diff[2 * j + 1]); });
                // ---- gradient of the maximin distance at the hotspot ----------
                //   -0.5 d(k12 / sqrt(k1 k2)) / dtheta / (d + 1e-4)
                // from the parked d k12 / d theta (_backend.cu:136-140,380-402).
                // F_REFCOMPAT reproduces the reference to the letter: its final
                // loop re-reads k12 -- and with it the distance in the
                // denominator -- from the solution buffer, which by then holds
                // the LAST perturbed solve (theta_last e^-eps, _backend.cu:383),
                // for every column from q on; the starting-probability columns
                // were finished before the finite-difference loop (:222-250).
                if constexpr (MAXIMIN) {
                    bool own = false;
#pragma unroll
                    for (int k = 0; k < R; ++k) own = own || mm_own[k];
                    if (own) {
                        const unsigned o1 = mm_hot / (unsigned)n2, o2 = mm_hot - o1 * (unsigned)n2;
                        const real k12_last = (mm_xlast - mm_corr) * mm_pp;
                        const real dd = real(0.9999995f) - k12_last * mm_rs;
                        const real D_last = real(sqrtf((float)(dd > real(0) ? dd : real(0))));
                        for (int col = 0; col < n_jac; ++col) {
                            const bool compat = (flags & F_REFCOMPAT) && col >= off_q;
                            const real k12u = compat ? k12_last : mm_k12;
                            const real Du = compat ? D_last : mm_D;
                            const size_t at = (size_t)I1 + (size_t)prm.nX * I2 + plane * col;
                            const real dk12 = prm.gradient[at];   // (this lane's own store)
                            const real dk1 = prm.diag_grad[(size_t)(mm_n1 + o1) + (size_t)prm.diag_ld * col];
                            const real dk2 = prm.diag_grad[(size_t)(mm_n2 + o2) + (size_t)prm.diag_ld * col];
                            const real dnorm = dk12 * mm_rs - real(0.5) * k12u * mm_rs * mm_rs * mm_rs *
                                               (dk1 * mm_k2 + mm_k1 * dk2);
                            const real g = real(-0.5) * dnorm / (Du + real(1e-4f));
                            prm.gradient[at] = g;
                            if (mirror) prm.gradient[(size_t)I2 + (size_t)prm.nX * I1 + plane * col] = g;
                        }
                    }
                }
            }

            // ---- analytic gradient (graph-level), marginalized_kernel.h:806-997
            if constexpr (C == 2) {
                job_sync<W>();
                if constexpr (SEQ) {
                    // two planes: lp[i] = YDq_i (the slots' gather addresses
                    // are those of the one-component solves), lp[cap + i] = Yp_i
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        lp[paddr[k]] = x[0][k];
                        lp[(unsigned)prm.u_capacity + (unsigned)paddr[k]] = x[1][k];
                    }
                } else {
                    publish(x);   // lp[2i] = YDq_i, lp[2i+1] = Yp_i
                }
                real jac[n_jac];
#pragma unroll
                for (int j = 0; j < n_jac; ++j) jac[j] = 0;
                const real Q = real(1) / (real(1) - q), Q3 = Q * Q * Q;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int pos = row_pos(k, wv, lane);
                    const bool ok = pos < N;
                    const unsigned rm = rowmap[ok ? pos : 0];
                    const int i1 = (int)(rm >> 16), i2 = (int)(rm & 0xFFFFu);
                    const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                    const real p1 = prm.p_start(v1), p2 = prm.p_start(v2);
                    const real dox = real(g1.degree[i1]) * real(g2.degree[i2]);
                    const real dx = dox * inv1q2;
                    const real v = kappa_v(i1, i2, v1, v2);
                    const real YDq = ok ? x[0][k] : real(0), Yp = ok ? x[1][k] : real(0);
                    auto dp1 = prm.p_start._j_a_c_o_b_i_a_n_(v1);
                    auto dp2 = prm.p_start._j_a_c_o_b_i_a_n_(v2);
#pragma unroll
                    for (int j = 0; j < PStart::jac_dims; ++j)
                        jac[j] += (real(dp1[j]) * p2 + p1 * real(dp2[j])) * YDq;
                    jac[off_q] += real(2) * Q * p1 * p2 * YDq - real(2) * Q3 * Yp * dox / v * YDq;
                    const real wv_ = dx * Yp * YDq / (v * v);
                    if constexpr (TAB) {
                        const unsigned cidx = __umul24((unsigned)ncls1[i1], nvc) + ncls2[i2];
#pragma unroll
                        for (int j = 0; j < NodeK::jac_dims; ++j)
                            jac[off_v + j] += wv_ * at32(dkvtab, (unsigned)j * nvc * nvc + cidx);
                    } else {
                        auto dv = prm.node_kernel._j_a_c_o_b_i_a_n_(v1, v2);
#pragma unroll
                        for (int j = 0; j < NodeK::jac_dims; ++j) jac[off_v + j] += wv_ * real(dv[j]);
                    }
                }
                job_sync<W>();
                if constexpr (EdgeK::jac_dims > 0 && FLY) {
                    // no slots: the owner of a row walks its terms once more,
                    // w = Yp(row) YDq(column) per term
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const unsigned rm = rowid[k];
                        const bool live = rm != ~0u;
                        const unsigned i1 = live ? rm >> 16 : 0u, i2 = live ? rm & 0xFFFFu : 0u;
                        const unsigned a0 = lrp1[i1], a1 = live ? (unsigned)lrp1[i1 + 1] : a0;
                        const unsigned b0 = lrp2[i2], b1 = live ? (unsigned)lrp2[i2 + 1] : b0;
                        const real yrow = x[1][k];
                        for (unsigned a = a0; a < a1; ++a) {
                            const edge_t e1 = at32(g1.edge, a);
                            const unsigned rowp = lp_off + __umul24((unsigned)at32(g1.nz, a).j, (unsigned)ldp) * ELEM;
                            [[maybe_unused]] unsigned c1 = 0;
                            real w1 = yrow;
                            if constexpr (TAB) {
                                c1 = __umul24((unsigned)ecls1[a], nec);
                                if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                    w1 *= real(edge_weight<edge_t>::get(e1));
                            }
                            for (unsigned b = b0; b < b1; ++b) {
                                const unsigned col = (unsigned)at32(g2.nz, b).j;
                                real w = w1 * load_real_at<real>(rowp + col * ELEM);
                                if constexpr (TAB) {
                                    const unsigned cidx = c1 + ecls2[b];
                                    if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                        w *= real(edge_weight<edge_t>::get(at32(g2.edge, b)));
#pragma unroll
                                    for (int jj = 0; jj < EdgeK::jac_dims; ++jj)
                                        jac[off_e + jj] += w * at32(dketab, (unsigned)jj * nec * nec + cidx);
                                } else {
                                    auto de = prm.edge_kernel._j_a_c_o_b_i_a_n_(e1, at32(g2.edge, b));
#pragma unroll
                                    for (int jj = 0; jj < EdgeK::jac_dims; ++jj) jac[off_e + jj] += w * real(de[jj]);
                                }
                            }
                        }
                    }
                } else if constexpr (EdgeK::jac_dims > 0) {
                    // walk the slots again: row = the batch's row, col = adr
                    int kb = 0;
                    walk_t cur = G0 ? walk_t{} : open_walk(0);
                    [[maybe_unused]] grid_t grid;
                    if constexpr (G0 > 0) {
                        grid = open_grid(0u, 0u);
                        cur.row = grid.row;
                    }
                    auto yp_of = [&](int row) -> real {
                        if constexpr (SEQ) return lp[(unsigned)prm.u_capacity + (unsigned)row];
                        else return lp[row * 2 + 1];
                    };
                    real yrow = yp_of(cur.row);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (s < n_slots) {   // wave-uniform
                            const bool ok = s < G0 ? grid.u[s / GD_] && grid.v[s % GD_] : cur.valid();
                            const unsigned a = s < G0 ? grid.a[s / GD_] : cur.a();
                            const unsigned b = s < G0 ? grid.b[s % GD_] : cur.e2;
                            real w = ok ? yrow * load_real_at<real>(adr[s]) : real(0);
                            if constexpr (TAB) {
                                const unsigned cidx = __umul24((unsigned)ecls1[a], nec) + ecls2[b];
                                if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                                    w *= real(edge_weight<edge_t>::get(g1.edge[a])) *
                                         real(edge_weight<edge_t>::get(g2.edge[b]));
#pragma unroll
                                for (int jj = 0; jj < EdgeK::jac_dims; ++jj)
                                    jac[off_e + jj] += w * at32(dketab, (unsigned)jj * nec * nec + cidx);
                            } else {
                                const edge_t e1 = g1.edge[a], e2 = g2.edge[b];
                                auto de = prm.edge_kernel._j_a_c_o_b_i_a_n_(e1, e2);
#pragma unroll
                                for (int jj = 0; jj < EdgeK::jac_dims; ++jj) jac[off_e + jj] += w * real(de[jj]);
                            }
                            if (s >= G0) cur.next();
                            if (flush_at(s, fm) && s != S - 1) {   // wave-uniform
                                ++kb;
                                cur = open_walk(kb);
                                yrow = yp_of(cur.row);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < n_jac; ++j) {
                    const real g = reduce::sum(jac[j], (j & 1) ? red1 : red0);
                    if (tid == 0) {
                        if (flags & F_PACKED) {
                            prm.gradient[(size_t)prm.order[t] * n_jac + j] = g;
                        } else if (flags & F_DIAGONAL) {
                            prm.gradient[(size_t)I1 + (size_t)prm.nX * j] = g;
                        } else {
                            const size_t plane = (size_t)prm.nX * prm.nY;
                            prm.gradient[(size_t)I1 + (size_t)prm.nX * I2 + plane * j] = g;
                            if (mirror) prm.gradient[(size_t)I2 + (size_t)prm.nX * I1 + plane * j] = g;
                        }
                    }
                }
            }
        }
    }

    // the first array if B, else the second (their extents differ)
    template<bool B, class X, class Y> __device__ static __forceinline__ auto &select_array(X &x, Y &y) {
        if constexpr (B) return x;
        else return y;
    }
    // lane LANE of v = s (s wave-uniform)
    // (readfirstlane: the "s" constraint alone does not make a value scalar --
    // where the compiler keeps the count in a vector register, as in the
    // nodal-gradient kernel of a rational-quadratic composite, it printed
    // `v_writelane_b32 v21, v39, 39`, which does not assemble: the JIT failed
    // at run time, scripts/fuzz_parity.py --modes=nodalgrad seed 122)
    template<int LANE> __device__ static __forceinline__ void writelane(int &v, int s) {
        asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(s)), "n"(LANE));
    }
    // class tables in the lanes of one register: lane c = first node of degree
    // class c in graph 1, lane 16 + c = nodes of class c in graph 2, lane
    // 32 + c = its first node (classes in descending degree: class DMAX first)
    template<int... Cs> __device__ static __forceinline__ void
    fill_classes(std::integer_sequence<int, Cs...>, int &vcls, const int (&cnt1)[NC], const int (&cnt2)[NC]) {
        int s1 = 0, s2 = 0;
        ((writelane<DMAX - Cs>(vcls, s1), writelane<16 + DMAX - Cs>(vcls, cnt2[DMAX - Cs]),
          writelane<32 + DMAX - Cs>(vcls, s2), s1 += cnt1[DMAX - Cs], s2 += cnt2[DMAX - Cs]), ...);
    }
    // sorted-row offset of rectangle (d1, d2) in lane (d1 * NC + d2) % 64 of
    // voff[(d1 * NC + d2) / 64]: a running sum in the compile-time order ORD
    template<int K, int NW> __device__ static __forceinline__ void write_offset(int (&voff)[NW], int off) {
        constexpr int idx = ORD.d1[K] * NC + ORD.d2[K];
        writelane<idx % 64>(voff[idx / 64], off);
    }
    template<int... Ks, int NW> __device__ static __forceinline__ void
    fill_offsets(std::integer_sequence<int, Ks...>, int (&voff)[NW], const int (&cnt1)[NC], const int (&cnt2)[NC]) {
        int off = 0;
        ((write_offset<Ks>(voff, off), off += cnt1[ORD.d1[Ks]] * cnt2[ORD.d2[Ks]]), ...);
    }
};

}  // namespace mgk
}  // namespace graphdot
#endif
