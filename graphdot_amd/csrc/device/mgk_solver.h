// Marginalized-graph-kernel pair solver for MI355X (gfx950, wave64).
//
// For a pair of graphs (G1, G2) with n1, n2 nodes the kernel value is
//   K = sum_i p1(i1) p2(i2) x_i,   (Dx Vx^-1 - Ax o Ex) x = Dx q^2/q0^2,
// solved by Jacobi-preconditioned CG on the N = n1*n2 product graph.  This is
// the computation of the reference's
//   graphdot/cpp/marginalized_kernel.h:189-490 (compute), :492-804
//   (compute_duo), :806-997 (derivative) and
//   graphdot/kernel/marginalized/template.cu:57-474 (job loop, lmin, output),
// re-designed for CDNA4 rather than translated:
//
//  * one wavefront (W = 1) or one workgroup of W wavefronts owns one pair;
//  * the product-graph operator is *materialised once per pair in registers*:
//    kappa_e(e1, e2) * w1 * w2 is evaluated once per nonzero, not once per
//    nonzero per CG iteration as the reference does, and kept with its gather
//    address in S register slots per lane; each lane also keeps the R rows of
//    x, r, p and the Jacobi diagonal it owns;
//  * the only per-iteration memory traffic is LDS: p is published to LDS and
//    the Kronecker mat-vec runs as two atomic-free stages
//        U[a, i2]    = sum_{b in adj(i2)} E[a, b] p[j1(a), j2(b)]    (stage 1)
//        (Wp)[i1,i2] = sum_{a in adj(i1)} U[a, i2]                    (stage 2)
//    (a, b index directed nonzeros of G1, G2).  A lane owns whole tasks
//    (a, i2) and whole rows (i1, i2), so sums accumulate in registers and
//    are written once -- no ds_add_f32, whose throughput on gfx950 turned out
//    to saturate the LDS pipe (profiles/r01_v1_atomic_scatter_pmc.csv);
//  * nodes are renumbered by descending degree (packer), tasks/rows are dealt
//    to lanes in that order, so the 64 tasks of one wave instruction have
//    (nearly) the same trip count and the wave-uniform maximum is the trip
//    count of its first task: padding is a few percent and *all control flow
//    in the iteration is wave-uniform* (a 64-bit flush mask, scalar loops);
//  * dot products are wave reductions (DPP + readlane), W > 1 adds one LDS
//    exchange per reduction;
//  * jobs are statically strided over the resident waves/workgroups in cost
//    order -- no global atomic job counter (reference: template.cu:58).
//
// Iteration structure, start vectors, preconditioner and the three stopping
// rules (rTz == 0, pAp == 0, sqrt(rTr) < tol*N, k == N) are the reference's.
#ifndef GRAPHDOT_HIP_MGK_SOLVER_H_
#define GRAPHDOT_HIP_MGK_SOLVER_H_
#include <hip/hip_runtime.h>
#include <utility>
#include "array.h"
#include "fmath.h"
#include "graph.h"
#include "numpy_type.h"
#include "wave.h"

namespace graphdot {
namespace mgk {

enum : unsigned {
    F_NODAL = 1u,      // write x_i p1 p2 per node pair instead of the sum
    F_DIAGONAL = 2u,   // jobs are (i, i); 1-D output
    F_SYMMETRIC = 4u,  // mirror K(I2, I1)
    F_LMIN1 = 8u,      // subtract the zero-step term kappa_v q^2/q0^2
    F_BLOCK = 16u,     // diag(nodal='block'): n x n block per graph
    F_PACKED = 32u,    // graph-level: out[job id] instead of K(I1, I2)
    F_REFCOMPAT = 64u, // maximin gradient: k12 of the last perturbed solve (mgk_oc.h)
    F_DENSE = 128u,    // on-the-fly launches: LDS holds room for the dense edge arrays (mgk_oc.h DENSE)
};

struct job_t {
    std::uint32_t i, j;
};

// Kernel parameter block (one by-value kernel argument; packed on the host
// with the same layout by _backend_hip.py).  Microkernel hyperparameters
// travel here (SGPR-resident kernargs) instead of __constant__ symbols
// (reference: _backend_cuda.py:318-340).
template<class real, class Graph, class NodeK, class EdgeK, class PStart> struct params_t {
    char const *arena;           // base of the graph arena (headers first)
    job_t const *jobs;
    std::uint32_t const *order;  // job ids of this launch, cost-descending
    std::uint32_t const *starts;
    real *gramian;
    real *gradient;
    std::uint32_t *iters;        // optional per-job CG iteration counts
    real *scratch;               // general / streamed solver: CG vectors in global memory
    unsigned *sync;              // streamed solver, several workgroups per pair: barrier cells and partial sums (mgk_stream.h)
    real *tables;                // microkernel values over label-class pairs (mgk_oc.h)
    // maximin distance (mgk_oc.h, MAXIMIN): nodal self-similarities of every
    // graph [node_starts[g] + node], their Jacobian (column-major, leading
    // dimension = total node count), the hotspot output, node offsets
    real const *diag;
    real const *diag_grad;
    std::int32_t *hotspot;
    std::uint32_t const *node_starts;
    std::uint32_t diag_ld;       // leading dimension of diag_grad (total node count)
    std::uint32_t n_launch_jobs;
    std::uint32_t nX, nY, nJ;
    std::uint32_t flags;
    std::uint32_t order_offset;  // slot of order[0] in the packed output
    std::uint32_t u_capacity;    // tasks per pair slot in the dynamic LDS region
    std::uint32_t g_capacity;    // bytes per staged graph image in dynamic LDS
    std::uint32_t parts;         // streamed solver: workgroups per pair (1: none of the grid-wide machinery)
    // label classes (GraphArena): counts and arena offsets of the class
    // representatives node_t[n_vclass], edge_t[n_eclass]
    std::uint32_t n_vclass, n_eclass, vrep, erep;
    real q, q0, eps, ftol, gtol;
    NodeK node_kernel;
    EdgeK edge_kernel;
    PStart p_start;
};

template<class real, int W> struct block_reduce {
    // Sum over all 64*W threads; `scratch` holds 2*W reals.
    __device__ static __forceinline__ void sum2(real &a, real &b, real *scratch) {
        wave::sum2(a, b);
        if constexpr (W > 1) {
            const int w = threadIdx.x / 64;
            __syncthreads();  // previous readers of scratch are done
            if (wave::laneid() == 0) {
                scratch[2 * w] = a;
                scratch[2 * w + 1] = b;
            }
            __syncthreads();
            real sa = 0, sb = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                sa += scratch[2 * k];
                sb += scratch[2 * k + 1];
            }
            a = sa;
            b = sb;
        }
    }
    __device__ static __forceinline__ real sum(real a, real *scratch) {
        a = wave::sum(a);
        if constexpr (W > 1) {
            const int w = threadIdx.x / 64;
            __syncthreads();
            if (wave::laneid() == 0) scratch[2 * w] = a;
            __syncthreads();
            real sa = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) sa += scratch[2 * k];
            a = sa;
        }
        return a;
    }
};

// Diagnostic build (-DGD_STAMPS): per-phase cycle totals are accumulated into
// prm.iters[n_jobs_total .. +4) -- never compiled into the product kernels.
#ifdef GD_STAMPS
#define GD_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define GD_STAMP(var)
#endif

template<int W> __device__ __forceinline__ void job_sync() {
    if constexpr (W > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
}

template<class real> __device__ __forceinline__ void lds_add(real *p, real v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// base[idx] with the byte offset formed in 32 bits: lets the compiler address
// with a scalar base + 32-bit vector offset (no 64-bit multiply-add per load)
template<class T> __device__ __forceinline__ T const &at32(T const *base, unsigned idx) {
    return *reinterpret_cast<T const *>(reinterpret_cast<char const *>(base) +
                                        idx * (unsigned)sizeof(T));
}

// Walks (hi, lo) = divmod(index, n) for index = start, start + step, ...
// without a division per step.
struct divmod_walk {
    int hi, lo, qs, rs, n;
    __device__ __forceinline__ divmod_walk(int start, int step, int n_) : n(n_) {
        // start, step < 2^22: float reciprocal + fix-up is exact
        const float inv = 1.0f / (float)n;
        hi = (int)((float)start * inv);
        hi -= (hi * n > start);
        hi += ((hi + 1) * n <= start);
        lo = start - hi * n;
        qs = (int)((float)step * inv);
        qs -= (qs * n > step);
        qs += ((qs + 1) * n <= step);
        rs = step - qs * n;
    }
    __device__ __forceinline__ void next() {
        hi += qs;
        lo += rs;
        if (lo >= n) {
            lo -= n;
            ++hi;
        }
    }
};

// LDS byte offset of a pointer into __shared__ memory
template<class T> __device__ __forceinline__ unsigned lds_offset(T *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) T *)p;
}

// The C interleaved components of element `i` (C = 2: one 8- or 16-byte
// vector access instead of two scalar ones -- `ds_read2_b32` of an element
// pair banks like two ds_read_b32 whose addresses are all even: a guaranteed
// 2-way conflict; `ds_read_b64` uses the 64-bank mapping).  `base` must be
// aligned to C reals.
template<int C, class real> __device__ __forceinline__ void load_elem(real const *base, unsigned i, real (&out)[C]) {
    if constexpr (C == 2) {
        typedef real vec2 __attribute__((ext_vector_type(2)));
        const vec2 v = reinterpret_cast<vec2 const *>(base)[i];
        out[0] = v.x;
        out[1] = v.y;
    } else {
        out[0] = base[i];
    }
}
template<int C, class real> __device__ __forceinline__ void store_elem(real *base, unsigned i, real const (&in)[C]) {
    if constexpr (C == 2) {
        typedef real vec2 __attribute__((ext_vector_type(2)));
        vec2 v;
        v.x = in[0];
        v.y = in[1];
        reinterpret_cast<vec2 *>(base)[i] = v;
    } else {
        base[i] = in[0];
    }
}

// The same through an LDS *byte address* kept in a register (a gather slot
// then costs one register and no address arithmetic inside the CG loop,
// whatever the register allocator would have liked to rematerialise).
template<int C, class real> __device__ __forceinline__ void load_elem_at(unsigned addr, real (&out)[C]) {
    if constexpr (C == 2) {
        typedef real vec2 __attribute__((ext_vector_type(2)));
        const vec2 v = *(const __attribute__((address_space(3))) vec2 *)addr;
        out[0] = v.x;
        out[1] = v.y;
    } else {
        out[0] = *(const __attribute__((address_space(3))) real *)addr;
    }
}
template<class real> __device__ __forceinline__ real load_real_at(unsigned addr) {
    return *(const __attribute__((address_space(3))) real *)addr;
}

// base[lane] = v for the 64 lanes of a wave, `base` wave-uniform: the store
// takes its address from M0 + 4 lane (ds_write_addtid_b32), i.e. no address
// VGPR and half the LDS cycles of ds_write_b32 (MI355X_MICROARCH.md, LDS).
// The compiler does not count this store in lgkmcnt, which can only make its
// own waits stricter; it is used where only the issuing wave reads the data
// back (LDS operations of one wave execute in order).
template<int BYTE_OFFSET> __device__ __forceinline__ void store_lane_contiguous(unsigned base, float v) {
    // (s_nop: one wait state between an SALU write of M0 and the add-TID store)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%2"
                 :: "s"(__builtin_amdgcn_readfirstlane((int)base)), "v"(v), "n"(BYTE_OFFSET) : "memory", "m0");
}

// sum_{j < n, j < deg} t[j] per lane, n <= 4: the entries past a lane's own
// degree hold other rows' data and are kept out by the EXEC mask, which only
// shrinks as j grows (v_cmpx writes it): two VALU per entry.
template<int n> __device__ __forceinline__ float masked_sum(float const (&t)[4], int deg) {
    float a;
    unsigned long long saved;
    if constexpr (n == 1)
        asm volatile("v_cmp_lt_i32 vcc, 0, %[dg]\n\tv_cndmask_b32 %[a], 0, %[t0], vcc"
                     : [a] "=&v"(a) : [dg] "v"(deg), [t0] "v"(t[0]) : "vcc");
    else if constexpr (n == 2)
        asm volatile("v_cmp_lt_i32 vcc, 0, %[dg]\n\tv_cndmask_b32 %[a], 0, %[t0], vcc\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32 vcc, 1, %[dg]\n\tv_add_f32 %[a], %[a], %[t1]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "=&v"(a), [sv] "=&s"(saved)
                     : [dg] "v"(deg), [t0] "v"(t[0]), [t1] "v"(t[1]) : "vcc");
    else if constexpr (n == 3)
        asm volatile("v_cmp_lt_i32 vcc, 0, %[dg]\n\tv_cndmask_b32 %[a], 0, %[t0], vcc\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32 vcc, 1, %[dg]\n\tv_add_f32 %[a], %[a], %[t1]\n\t"
                     "v_cmpx_lt_i32 vcc, 2, %[dg]\n\tv_add_f32 %[a], %[a], %[t2]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "=&v"(a), [sv] "=&s"(saved)
                     : [dg] "v"(deg), [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]) : "vcc");
    else
        asm volatile("v_cmp_lt_i32 vcc, 0, %[dg]\n\tv_cndmask_b32 %[a], 0, %[t0], vcc\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32 vcc, 1, %[dg]\n\tv_add_f32 %[a], %[a], %[t1]\n\t"
                     "v_cmpx_lt_i32 vcc, 2, %[dg]\n\tv_add_f32 %[a], %[a], %[t2]\n\t"
                     "v_cmpx_lt_i32 vcc, 3, %[dg]\n\tv_add_f32 %[a], %[a], %[t3]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "=&v"(a), [sv] "=&s"(saved)
                     : [dg] "v"(deg), [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]), [t3] "v"(t[3])
                     : "vcc");
    return a;
}
template<int n> __device__ __forceinline__ double masked_sum(double const (&t)[4], int deg) {
    static_assert(n == 4, "the double build only instantiates the 4-entry sum");
    double a = 0;
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_lt_i32 vcc, 0, %[dg]\n\tv_add_f64 %[a], %[a], %[t0]\n\t"
                 "v_cmpx_lt_i32 vcc, 1, %[dg]\n\tv_add_f64 %[a], %[a], %[t1]\n\t"
                 "v_cmpx_lt_i32 vcc, 2, %[dg]\n\tv_add_f64 %[a], %[a], %[t2]\n\t"
                 "v_cmpx_lt_i32 vcc, 3, %[dg]\n\tv_add_f64 %[a], %[a], %[t3]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [a] "+v"(a), [sv] "=&s"(saved)
                 : [dg] "v"(deg), [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]), [t3] "v"(t[3])
                 : "vcc");
    return a;
}

// S: register slots per lane for stage-1 nonzeros, R: rows per lane,
// W: waves per pair, C: right-hand sides (1 = value, 2 = value + gradient),
// NODAL: compile the node-wise output modes (F_NODAL / F_BLOCK) in.
// Does the edge record carry a weight (edge_t = {weight, label} for weighted
// graphs, _devicegraph.py; the translation unit says so with GD_WEIGHTED)?
// The class tables hold the label part only.
#ifndef GD_WEIGHTED
#define GD_WEIGHTED 0
#endif
template<class E, class = void> struct edge_weight {
    constexpr static bool value = false;
    __device__ static float get(E const &) { return 1.f; }
};
template<class E> struct edge_weight<E, decltype(void(std::declval<E const &>().weight))> {
    constexpr static bool value = true;
    __device__ static auto get(E const &e) { return e.weight; }
};

// TAB: the labels of this call fall into few classes (GraphArena.classes):
//   every workgroup evaluates the node and edge microkernels once per class
//   pair into an LDS table, and slot / row setup look values up by the u8
//   class ids staged with the graph images instead of evaluating the
//   microkernels per nonzero pair and per row.
template<class real, int S, int R, int W, int C, bool NODAL, bool TAB, class Graph, class NodeK, class EdgeK, class PStart>
struct pair_solver {
    using P = params_t<real, Graph, NodeK, EdgeK, PStart>;
    using node_t = typename Graph::node_t;
    using edge_t = typename Graph::edge_t;
    constexpr static int T = 64 * W;   // threads per pair
    constexpr static int NV = R * T;   // capacity of p (rows)
    constexpr static int WPB = 1;                 // independent pairs per workgroup (four one-wave pairs per 256-thread workgroup: 4-5 % slower, round 1)
    constexpr static int threads = 64 * W * WPB;
    constexpr static int NM = (S + 31) / 32;      // 32-bit flush-mask words
    constexpr static int SETUP_CHUNK = 4;
    constexpr static int GCH = 8;                 // stage-1 gathers in flight
    constexpr static int ZPAD = 64;               // zero entries at the end of U
    constexpr static int DU = 4;                  // stage-2 degree bound of the unrolled path
    constexpr static int RCH = 2;                 // rows whose stage-2 reads are in flight together
    // one wave per pair, one float per entry: stores of U and p go through M0
    constexpr static bool ADDTID = W == 1 && C == 1 && sizeof(real) == 4;
    constexpr static int n_jac = PStart::jac_dims + 1 + NodeK::jac_dims + EdgeK::jac_dims;
    constexpr static int off_q = PStart::jac_dims;
    constexpr static int off_v = off_q + 1;
    constexpr static int off_e = off_v + NodeK::jac_dims;

    // p and the reduction scratch are static LDS; U (one entry per stage-1
    // task, at most u_capacity per pair) lives in the dynamic LDS region,
    // sized per launch from the largest pair in it.
    struct lds_t {
        alignas(16) real p[WPB][NV * C];
        real red[WPB][2 * W];
    };

    // wave-uniform integer (kept in an SGPR)
    __device__ static __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

    // stage-1 task of lane `tid` in batch kb: task index kb*T + tid = i2*ldu + a
    struct task_t {
        bool ok;
        int a, i2, rs2, deg;
    };

    // lp[row * C + c] = p[c][k] for the rows k*T + tid of this thread
    template<int k = 0>
    __device__ static __forceinline__ void publish_p(real *lp, int tid, real const (&p)[C][R]) {
        if constexpr (ADDTID) {
            if constexpr (k < R) {
                store_lane_contiguous<k * T * 4>(lds_offset(lp), (float)p[0][k]);
                publish_p<k + 1>(lp, tid, p);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < R; ++kk) {
                real e[C];
#pragma unroll
                for (int c = 0; c < C; ++c) e[c] = p[c][kk];
                store_elem<C>(lp, kk * T + tid, e);
            }
        }
    }

    __device__ static __forceinline__ void run(P const &prm, lds_t &lds, real *dyn) {
        const int lane = wave::laneid();
        const int slot = (W == 1) ? uni((int)(threadIdx.x / 64)) : 0;  // pair slot in workgroup
        const int tid = (W == 1) ? lane : (int)threadIdx.x;             // thread within pair
        const int wv = (W == 1) ? 0 : uni((int)(threadIdx.x / 64));     // wave within pair
        real *const lp = lds.p[slot];
        // dynamic LDS per pair slot: [U: u_capacity*C reals][G1 image][G2 image]
        const unsigned slot_bytes = prm.u_capacity * C * (unsigned)sizeof(real) + 2 * prm.g_capacity;
        char *const dyn_slot = reinterpret_cast<char *>(dyn) + (size_t)slot * slot_bytes;
        real *const lU = reinterpret_cast<real *>(dyn_slot);
        char *const lG1 = dyn_slot + prm.u_capacity * C * sizeof(real);
        char *const lG2 = lG1 + prm.g_capacity;
        real *const red = lds.red[slot];
        const unsigned lU_off = uni((int)lds_offset(lU));
        graph_header_t const *const headers = reinterpret_cast<graph_header_t const *>(prm.arena);
        // microkernel tables, shared by the pair slots of the workgroup:
        // [kv: n_vclass^2][ke: n_eclass^2] behind the per-slot regions
        real *const kvtab = reinterpret_cast<real *>(reinterpret_cast<char *>(dyn) + (size_t)WPB * slot_bytes);
        real *const ketab = kvtab + prm.n_vclass * prm.n_vclass;
        if constexpr (TAB) {
            node_t const *const vr = reinterpret_cast<node_t const *>(prm.arena + prm.vrep);
            edge_t const *const er = reinterpret_cast<edge_t const *>(prm.arena + prm.erep);
            const unsigned nv = prm.n_vclass, ne = prm.n_eclass;
            for (unsigned k = threadIdx.x; k < nv * nv; k += threads) {
                const unsigned c1 = k / nv, c2 = k - c1 * nv;
                kvtab[k] = real(prm.node_kernel(vr[c1], vr[c2]));
            }
            for (unsigned k = threadIdx.x; k < ne * ne; k += threads) {
                const unsigned c1 = k / ne, c2 = k - c1 * ne;
                ketab[k] = real(prm.edge_kernel(er[c1], er[c2]));   // weights of the representatives are 1
            }
            __syncthreads();
        }

#ifdef GD_STAMPS
        unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
        const unsigned n_units = gridDim.x * WPB;
        for (unsigned t = blockIdx.x * WPB + slot; t < prm.n_launch_jobs; t += n_units) {
            GD_STAMP(t_begin);
            const job_t job = prm.jobs[t];     // jobs of this launch, in launch order
            const graph_header_t h1 = headers[job.i], h2 = headers[job.j];
            const int n1 = h1.n_node, n2 = h2.n_node, N = n1 * n2;
            const int nnz1 = h1.n_nz;
            // LDS strides are odd so that the per-lane addresses of one access
            // spread over the 32 banks (n2 and nnz1 are even more often than
            // not: 16 x 16 nodes put all rows on 2 banks, 32 nonzeros all rows'
            // U entries on one).  p: row i1 starts at i1*ldp; U: the tasks of
            // i2 start at i2*ldu.  Both index spaces are dealt to lanes padded
            // -- the pad row (i1, n2) and pad task (nnz1, i2) are dead lanes
            // (+3..6 % lanes) -- so that every store stays lane-contiguous.
            const int ldp = n2 | 1;
            const int ldu = nnz1 + 1;          // nnz1 = 2 * edges
            const int NP = n1 * ldp;           // padded row space
            const int ntask = ldu * n2;        // stage-1 tasks (a, i2), i2-major, padded
            const real q = prm.q, q0 = prm.q0;
            const real inv1q2 = real(1) / ((real(1) - q) * (real(1) - q));
            const real bscale = q * q / (q0 * q0);
            // label-class section in front of each image (_devicegraph.class_bytes):
            // u8 node classes [pad4(n)], u8 edge classes [pad4(nnz)], 16-aligned
            const unsigned nzpad1 = ((unsigned)n1 + 3u) & ~3u, nzpad2 = ((unsigned)n2 + 3u) & ~3u;
            const unsigned cb1 = TAB ? (nzpad1 + (((unsigned)nnz1 + 3u) & ~3u) + 15u) & ~15u : 0u;
            const unsigned cb2 = TAB ? (nzpad2 + (((unsigned)h2.n_nz + 3u) & ~3u) + 15u) & ~15u : 0u;

            // ---- stage both graph images in LDS: one global round trip --------
            // A packed image is contiguous ([degree .. perm], graph.h); every
            // later label / CSR access is an LDS read.  (Variable-length
            // attribute payloads stay in global memory behind their pointers.)
            job_sync<W>();  // previous pair is done with lU / lG
            {
                // 16 bytes per lane and load: every section (and the class
                // section in front) is 16-byte aligned in the arena, the LDS
                // images are too, and g_capacity is a multiple of 16
                typedef unsigned v4 __attribute__((ext_vector_type(4)));
                const unsigned w1 = (h1.perm + 2u * n1 - h1.degree + cb1 + 15u) / 16u;
                const unsigned w2 = (h2.perm + 2u * n2 - h2.degree + cb2 + 15u) / 16u;
                const v4 *const s1 = reinterpret_cast<const v4 *>(prm.arena + h1.degree - cb1);
                const v4 *const s2 = reinterpret_cast<const v4 *>(prm.arena + h2.degree - cb2);
                v4 *const d1 = reinterpret_cast<v4 *>(lG1);
                v4 *const d2 = reinterpret_cast<v4 *>(lG2);
                constexpr int K = 2;   // loads in flight per graph and lane
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
            }
            // graph views whose section pointers land in the LDS images
            const Graph g1(lG1 + cb1 - h1.degree, h1);
            const Graph g2(lG2 + cb2 - h2.degree, h2);
            std::uint8_t const *const ncls1 = reinterpret_cast<std::uint8_t const *>(lG1);
            std::uint8_t const *const ncls2 = reinterpret_cast<std::uint8_t const *>(lG2);
            std::uint8_t const *const ecls1 = ncls1 + nzpad1;
            std::uint8_t const *const ecls2 = ncls2 + nzpad2;
            // microkernel value of a node / edge pair: table or direct
            auto kappa_v = [&](int i1, int i2, node_t const &v1, node_t const &v2) -> real {
                if constexpr (TAB) return kvtab[__umul24((unsigned)ncls1[i1], prm.n_vclass) + ncls2[i2]];
                else return real(prm.node_kernel(v1, v2));
            };
            std::uint16_t const *const lrp1 = g1.rowptr;
            std::uint16_t const *const lrp2 = g2.rowptr;
            // the last ZPAD entries of the U region stay zero: rows past N
            // point there, so stage 2 can read without a per-lane mask
            const int zbase = (int)prm.u_capacity - ZPAD;
            if (tid < ZPAD) {
#pragma unroll
                for (int c = 0; c < C; ++c) lU[(zbase + tid) * C + c] = 0;
            }
            job_sync<W>();
            GD_STAMP(t_stage);

            // ---- stage-1 nonzeros owned by this thread ------------------------
            // batch kb = tasks [kb*T, kb*T + T); this lane's task is kb*T + tid.
            // All lanes of a wave walk D = deg2(first task of the wave) slots
            // per batch; the slot after which a batch ends is marked in `fm`.
            // Pass 1 (index walk, needs only the CSR row pointers of G2 which
            // were staged in LDS) records per slot the nonzero pair (a, b);
            // pass 2 then issues all label loads back to back and evaluates
            // the edge kernel.
            real val[S];
            unsigned adr[S];   // pass 1: (a << 16) | b, or ~0u; pass 2: gather index into p
            unsigned fm[NM];
#pragma unroll
            for (int w = 0; w < NM; ++w) fm[w] = 0;
            int n_slots = 0;
            {
                // pass 0 (wave-uniform, rolled): depth of every batch of this
                // wave -> flush mask and slot count
                {
                    divmod_walk tf(64 * wv, T, ldu);       // first task of the wave
#pragma nounroll
                    for (int f = 64 * wv; f < ntask; f += T) {
                        const int fi2 = uni(tf.hi);
                        int D = uni(lrp2[fi2 + 1] - lrp2[fi2]);
                        D = D < 1 ? 1 : D;
                        n_slots += D;
                        const int last = n_slots - 1;      // < S (host guarantees)
#pragma unroll
                        for (int w = 0; w < NM; ++w)
                            if (last / 32 == w) fm[w] |= 1u << (last % 32);
                        tf.next();
                    }
                    n_slots = n_slots > S ? S : n_slots;
                }
                // pass 1 (unrolled, registers only + LDS row pointers): the
                // nonzero pair of every slot of this lane
                divmod_walk tk(tid, T, ldu);           // per lane: (i2, a)
                int kb = 0, d = 0;
                auto open_task = [&]() -> task_t {
                    task_t k;
                    k.ok = kb * T + tid < ntask && tk.lo < nnz1;
                    k.i2 = k.ok ? tk.hi : 0;
                    k.a = k.ok ? tk.lo : 0;
                    k.rs2 = lrp2[k.i2];
                    k.deg = k.ok ? lrp2[k.i2 + 1] - k.rs2 : 0;
                    return k;
                };
                task_t cur = open_task();
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    adr[s] = (s < n_slots && cur.ok && d < cur.deg)
                        ? ((unsigned)cur.a << 16) | (unsigned)(cur.rs2 + d) : ~0u;
                    // materialise now: otherwise the select is sunk into pass 2
                    // and a snapshot of the task state stays live per slot
                    asm volatile("" : "+v"(adr[s]));
                    ++d;
                    if ((fm[s / 32] >> (s % 32)) & 1u) {   // wave-uniform
                        ++kb;
                        tk.next();
                        d = 0;
                        cur = open_task();
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    // bound the loads in flight (and with them the live
                    // registers) to SETUP_CHUNK slots
                    if (s % SETUP_CHUNK == 0) __builtin_amdgcn_sched_barrier(0);
                    const bool ok = adr[s] != ~0u;
                    const unsigned a = ok ? (adr[s] >> 16) : 0u, b = ok ? (adr[s] & 0xFFFFu) : 0u;
                    const nz_t z1 = at32(g1.nz, a), z2 = at32(g2.nz, b);
                    real e;
                    if constexpr (TAB) {
                        e = ketab[__umul24((unsigned)ecls1[a], prm.n_eclass) + ecls2[b]];
                        if constexpr (GD_WEIGHTED && edge_weight<edge_t>::value)
                            e *= real(edge_weight<edge_t>::get(at32(g1.edge, a))) *
                                 real(edge_weight<edge_t>::get(at32(g2.edge, b)));
                    } else {
                        const edge_t e1 = at32(g1.edge, a), e2 = at32(g2.edge, b);
                        e = prm.edge_kernel(e1, e2);
                    }
                    val[s] = ok ? e : real(0);
                    unsigned col = ok ? __umul24((unsigned)z1.j, (unsigned)ldp) + (unsigned)z2.j : 0u;
                    // pin the evaluation here: otherwise it is sunk below the
                    // last chunk and every slot's raw labels stay live
                    asm volatile("" : "+v"(val[s]), "+v"(col));
                    adr[s] = col;
                }
            }
            // (the compiler turns adr[s] into the LDS address &lp[adr[s] * C]
            // once, outside the CG loop: one register per slot, no per-
            // iteration address arithmetic)
            auto gather_index = [&](int s) -> unsigned { return adr[s]; };

            GD_STAMP(t_slots);
            // ---- rows owned by this thread ------------------------------------
            // row i = k*T + tid = (i1, i2); Jacobi diagonal, start vectors, and
            // for stage 2 the first task index / trip count of the row.
            real dg[R], mi[R], x[C][R], r[C][R], p[C][R];
            int ubase[R], udeg[R];
            int D1[R], D0[R];   // wave-uniform max / min degree of the rows of batch k
            int plain1[R];
            real rTz = 0;
            {
                divmod_walk row(tid, T, ldp);         // per lane
                divmod_walk first(64 * wv, T, ldp);   // first row of this wave (uniform)
                divmod_walk lastw(64 * wv + 63, T, ldp);   // its last row (uniform)
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const bool ok = k * T + tid < NP && row.lo < n2;
                    const int i1 = ok ? row.hi : 0, i2 = ok ? row.lo : 0;
                    const node_t v1 = at32(g1.node, (unsigned)i1), v2 = at32(g2.node, (unsigned)i2);
                    const real dx = real(at32(g1.degree, (unsigned)i1)) *
                                    real(at32(g2.degree, (unsigned)i2)) * inv1q2;
                    const real vx = kappa_v(i1, i2, v1, v2);
                    dg[k] = ok ? dx / vx : real(0);
                    mi[k] = ok ? vx / dx : real(0);
                    const int rs = lrp1[i1];
                    udeg[k] = ok ? lrp1[i1 + 1] - rs : 0;
                    // rows without neighbours (isolated nodes, dead lanes, pad
                    // rows) point at the zero pad: whatever stage 2 reads for
                    // them unmasked is zero
                    ubase[k] = udeg[k] > 0 ? (int)__umul24((unsigned)i2, (unsigned)ldu) + rs : zbase;
                    // rows of one wave are consecutive: the first has the largest
                    // degree, the last the smallest (0 if the batch has dead rows)
                    const int f1 = uni(first.hi);
                    D1[k] = (k * T + 64 * wv < NP) ? uni(lrp1[f1 + 1] - lrp1[f1]) : 0;
                    const int l1 = uni(lastw.hi);
                    D0[k] = (k * T + 64 * wv + 63 < NP) ? uni(lrp1[l1 + 1] - lrp1[l1]) : 0;
                    // (pad rows inside a batch read the zero pad unmasked)
                    D0[k] = D0[k] < ZPAD ? D0[k] : ZPAD;
                    // a batch of degree-1 rows (hydrogens; its dead lanes and pad
                    // rows read the zero pad) needs no mask in stage 2
                    plain1[k] = uni(D1[k] <= 1 && D0[k] == D1[k]);
                    asm volatile("" : "+s"(plain1[k]));   // an SGPR now, not a sunk readfirstlane
                    const real b = ok ? dx * bscale : real(0);
                    x[0][k] = 0;
                    r[0][k] = b;
                    p[0][k] = b * mi[k];
                    rTz += r[0][k] * p[0][k];
                    if constexpr (C == 2) {
                        const real bx = ok ? real(prm.p_start(v1)) * real(prm.p_start(v2)) : real(0);
                        x[1][k] = 0;
                        r[1][k] = bx;
                        p[1][k] = bx * mi[k];
                        rTz += r[1][k] * p[1][k];
                    }
                    row.next();
                    first.next();
                    lastw.next();
                }
            }

            GD_STAMP(t_setup);
            // ---- publish p ---------------------------------------------------
            job_sync<W>();  // everyone is done with the staged row pointers
            publish_p(lp, tid, p);
            rTz = block_reduce<real, W>::sum(rTz, red);

            const real tol = (C == 2) ? real(1e-10) * real(2 * N) : prm.ftol * real(N);
            const real tol2 = tol * tol;
            unsigned it = 0;
            for (; it < (unsigned)N && rTz != real(0); ++it) {
                job_sync<W>();   // p published
                // stage 1: U[task] = sum_b E[a, b] p[j1(a), j2(b)]
                {
                    // gathers are issued GCH at a time ahead of their use
                    // (padding slots read p[0]); the segmented sums flush at
                    // wave-uniform positions
                    real acc[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = 0;
                    int kb = 0;
                    // keep the flush mask as data (s_bitcmp per slot) instead of
                    // letting LICM expand it into S precomputed SGPR pairs
                    unsigned fmv[NM];
#pragma unroll
                    for (int w = 0; w < NM; ++w) {
                        fmv[w] = (unsigned)__builtin_amdgcn_readfirstlane((int)fm[w]);
                        asm volatile("" : "+s"(fmv[w]));
                    }
#pragma unroll
                    for (int s0 = 0; s0 < S; s0 += GCH) {
                        if (s0 >= n_slots) break;   // wave-uniform: no slots left
                        real g[C][GCH];
#pragma unroll
                        for (int j = 0; j < GCH; ++j) {
                            real e[C];
#pragma unroll
                            for (int c = 0; c < C; ++c) e[c] = 0;
                            if (s0 + j < S) load_elem<C>(lp, gather_index(s0 + j), e);
#pragma unroll
                            for (int c = 0; c < C; ++c) g[c][j] = e[c];
                        }
#pragma unroll
                        for (int j = 0; j < GCH; ++j) {
                            const int s = s0 + j;
                            if (s < S) {
#pragma unroll
                                for (int c = 0; c < C; ++c) acc[c] += val[s] * g[c][j];
                                if ((fmv[s / 32] >> (s % 32)) & 1u) {   // wave-uniform
                                    if constexpr (ADDTID) {
                                        store_lane_contiguous<0>(lU_off + kb * (T * 4), (float)acc[0]);
                                        acc[0] = 0;
                                    } else {
                                        store_elem<C>(lU, kb * T + tid, acc);
#pragma unroll
                                        for (int c = 0; c < C; ++c) acc[c] = 0;
                                    }
                                    ++kb;
                                }
                            }
                        }
                    }
                }
                job_sync<W>();
                // stage 2: Ap = diag.p - sum_{a in adj(i1)} U[a, i2]
                real Ap[C][R];
                real pAp = 0;
                {
                    real acc[C][R];
#pragma unroll
                    for (int k = 0; k < R; ++k)
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c][k] = 0;
                    // Rows are dealt in descending-degree order, so within batch k
                    // all 64 rows have D0[k] <= degree <= D1[k] (wave-uniform
                    // bounds).
                    int fast = uni(D1[0] <= DU);   // D1[0] bounds every degree this wave sees
                    asm volatile("" : "+s"(fast));   // one CG loop, not two unswitched copies
                    if (fast) {
                        // Degrees up to DU (every molecular graph): read DU entries
                        // of every row, back to back with immediate offsets and no
                        // branch -- what lies past a row's degree is some other
                        // row's entry or the zero pad -- then sum them under the
                        // lane's own degree as EXEC mask (masked_sum), or take the
                        // single entry as it is in a batch of degree-1 rows.  One
                        // LDS round trip per RCH rows instead of one per entry.
                        // (RCH rows at a time bounds the registers in flight)
#pragma unroll
                        for (int kc = 0; kc < R; kc += RCH) {
                            real u[RCH][C][DU];
#pragma unroll
                            for (int k = kc; k < kc + RCH && k < R; ++k) {
                                real const *const u0 = lU + ubase[k] * C;
#pragma unroll
                                for (int j = 0; j < DU; ++j) {
                                    real e[C];
                                    load_elem<C>(u0, j, e);
#pragma unroll
                                    for (int c = 0; c < C; ++c) u[k - kc][c][j] = e[c];
                                }
                            }
#pragma unroll
                            for (int k = kc; k < kc + RCH && k < R; ++k) {
                                // (kept as data: hoisted out of the CG loop the test
                                // would pin an SGPR pair per row)
                                int plain = __builtin_amdgcn_readfirstlane(plain1[k]);   // wave-uniform
                                asm volatile("" : "+s"(plain));
#pragma unroll
                                for (int c = 0; c < C; ++c) {
                                    real (&t)[DU] = u[k - kc][c];
                                    acc[c][k] = plain ? t[0] : masked_sum<4>(t, udeg[k]);
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            real const *const u0 = lU + ubase[k] * C;
                            int d = 0;
                            for (; d + 4 <= D0[k]; d += 4) {
                                real u[C][4];
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int c = 0; c < C; ++c) u[c][j] = u0[(d + j) * C + c];
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int c = 0; c < C; ++c) acc[c][k] += u[c][j];
                            }
                            for (; d < D0[k]; ++d)
#pragma unroll
                                for (int c = 0; c < C; ++c) acc[c][k] += u0[d * C + c];
                            for (; d < D1[k]; ++d) {
                                real const *const src = (d < udeg[k]) ? u0 + d * C : lU + zbase * C;
#pragma unroll
                                for (int c = 0; c < C; ++c) acc[c][k] += src[c];
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k)
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            Ap[c][k] = dg[k] * p[c][k] - acc[c][k];
                            pAp += p[c][k] * Ap[c][k];
                        }
                }
                pAp = block_reduce<real, W>::sum(pAp, red);
                if (pAp == real(0)) break;
                const real alpha = rTz / pAp;
                real rTr = 0, rTz_next = 0;
                real z[C][R];
#pragma unroll
                for (int k = 0; k < R; ++k)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        x[c][k] += alpha * p[c][k];
                        r[c][k] -= alpha * Ap[c][k];
                        z[c][k] = mi[k] * r[c][k];
                        rTr += r[c][k] * r[c][k];
                        rTz_next += r[c][k] * z[c][k];
                    }
                block_reduce<real, W>::sum2(rTr, rTz_next, red);
                if (rTr < tol2) {   // sqrt(rTr) < tol
                    ++it;
                    break;
                }
                const real beta = rTz_next / rTz;
#pragma unroll
                for (int k = 0; k < R; ++k)
#pragma unroll
                    for (int c = 0; c < C; ++c) p[c][k] = z[c][k] + beta * p[c][k];
                publish_p(lp, tid, p);
                rTz = rTz_next;
            }
            if (prm.iters != nullptr && tid == 0) prm.iters[prm.order[t]] = it;
            GD_STAMP(t_loop);

            // ---- output ------------------------------------------------------
            const unsigned flags = prm.flags;
            const unsigned I1 = prm.starts[job.i], I2 = prm.starts[job.j];
            const bool mirror = (flags & F_SYMMETRIC) && job.i != job.j;
            real ksum = 0;
            {
                divmod_walk row(tid, T, ldp);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const bool ok = k * T + tid < NP && row.lo < n2;
                    const int i1 = ok ? row.hi : 0, i2 = ok ? row.lo : 0;
                    row.next();
                    const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                    real xi = x[0][k];
                    if (flags & F_LMIN1) xi -= kappa_v(i1, i2, v1, v2) * bscale;
                    const real pp = real(prm.p_start(v1)) * real(prm.p_start(v2));
                    const real rv = ok ? xi * pp : real(0);
                    ksum += rv;
                    if constexpr (NODAL) if ((flags & F_NODAL) && ok) {
                        // back to the caller's node numbering
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
            if (!NODAL || !(flags & F_NODAL)) {
                ksum = block_reduce<real, W>::sum(ksum, red);
                if (tid == 0) {
                    if (flags & F_PACKED) {
                        prm.gramian[prm.order[t]] = ksum;   // slab in job order
                    } else if (flags & F_DIAGONAL) {
                        prm.gramian[I1] = ksum;
                    } else {
                        prm.gramian[(size_t)I1 + (size_t)prm.nX * I2] = ksum;
                        if (mirror) prm.gramian[(size_t)I2 + (size_t)prm.nX * I1] = ksum;
                    }
                }
            }

            // ---- analytic gradient (graph-level), marginalized_kernel.h:806-997
            if constexpr (C == 2) {
                // publish both solutions: lp[2i] = YDq_i, lp[2i+1] = Yp_i
                job_sync<W>();
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const real e[2] = {x[0][k], x[1][k]};
                    store_elem<2>(lp, k * T + tid, e);
                }
                real jac[n_jac];
#pragma unroll
                for (int j = 0; j < n_jac; ++j) jac[j] = 0;
                const real Q = real(1) / (real(1) - q), Q3 = Q * Q * Q;
                {
                    divmod_walk row(tid, T, ldp);
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const bool ok = k * T + tid < NP && row.lo < n2;
                        const int i1 = ok ? row.hi : 0, i2 = ok ? row.lo : 0;
                        row.next();
                        const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                        const real p1 = prm.p_start(v1), p2 = prm.p_start(v2);
                        const real dox = real(g1.degree[i1]) * real(g2.degree[i2]);
                        const real dx = dox * inv1q2;
                        const real v = prm.node_kernel(v1, v2);
                        const real YDq = ok ? x[0][k] : real(0), Yp = ok ? x[1][k] : real(0);
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
                }
                job_sync<W>();
                if constexpr (EdgeK::jac_dims > 0) {
                    // walk the stage-1 nonzeros again: row = (i1(a), i2), col = adr
                    divmod_walk tk(tid, T, ldu);
                    int kb = 0, d = 0;
                    auto open_task = [&]() -> task_t {
                        task_t k;
                        k.ok = kb * T + tid < ntask && tk.lo < nnz1;
                        k.i2 = k.ok ? tk.hi : 0;
                        k.a = k.ok ? tk.lo : 0;
                        k.rs2 = g2.rowptr[k.i2];
                        k.deg = k.ok ? (int)g2.rowptr[k.i2 + 1] - k.rs2 : 0;
                        return k;
                    };
                    task_t cur = open_task();
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (s < n_slots) {   // wave-uniform
                            const bool ok = cur.ok && d < cur.deg;
                            const int b = ok ? cur.rs2 + d : 0;
                            const edge_t e1 = g1.edge[cur.a], e2 = g2.edge[b];
                            auto de = prm.edge_kernel._j_a_c_o_b_i_a_n_(e1, e2);
                            const int row = g1.nz[cur.a].i * ldp + cur.i2;
                            const real w = ok ? lp[row * 2 + 1] * lp[gather_index(s) * 2 + 0] : real(0);
#pragma unroll
                            for (int j = 0; j < EdgeK::jac_dims; ++j) jac[off_e + j] += w * real(de[j]);
                            ++d;
                            if ((fm[s / 32] >> (s % 32)) & 1u) {   // wave-uniform
                                ++kb;
                                d = 0;
                                tk.next();
                                cur = open_task();
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < n_jac; ++j) {
                    const real g = block_reduce<real, W>::sum(jac[j], red);
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
#ifdef GD_STAMPS
            {
                const unsigned long long t_end = __builtin_amdgcn_s_memtime();
                st_acc[0] += t_setup - t_begin;
                st_acc[1] += t_loop - t_setup;
                st_acc[2] += t_end - t_loop;
                st_acc[3] += 1;
                st_acc[4] += t_stage - t_begin;
                st_acc[5] += t_slots - t_stage;
                st_acc[6] += t_setup - t_slots;
            }
#endif
        }
#ifdef GD_STAMPS
        if (tid == 0 && prm.iters != nullptr) {
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(
                prm.iters + ((prm.nX * prm.nY + 1) & ~1u));   // behind the per-job counters
            for (int k = 0; k < 7; ++k) atomicAdd(acc + k, st_acc[k]);
        }
#endif
    }
};


// ---------------------------------------------------------------------------
// General solver: any pair size.  One workgroup per pair; the CG vectors and
// the stage-1 products U live in a per-workgroup global-memory scratch
// (the role of the reference's PCGScratch, kernel/marginalized/_scratch.py:20-36,
// marginalized_kernel.h:20-38) and the edge kernel is evaluated on the fly in
// every iteration, like the reference does.  Same two-stage atomic-free
// mat-vec, same iteration and stopping rules, same outputs as pair_solver;
// used only for pairs that exceed the largest register-resident variant.
// scratch layout per workgroup: [x | r | p | U] with x, r, p of N*C reals and
// U of ntask*C reals (capacity passed in prm.u_capacity = reals per workgroup).
// ---------------------------------------------------------------------------
template<class real, int TPB, int C, class Graph, class NodeK, class EdgeK, class PStart>
struct general_solver {
    using P = params_t<real, Graph, NodeK, EdgeK, PStart>;
    using node_t = typename Graph::node_t;
    using edge_t = typename Graph::edge_t;
    constexpr static int W = TPB / 64;
    constexpr static int n_jac = PStart::jac_dims + 1 + NodeK::jac_dims + EdgeK::jac_dims;
    constexpr static int off_q = PStart::jac_dims;
    constexpr static int off_v = off_q + 1;
    constexpr static int off_e = off_v + NodeK::jac_dims;

    struct lds_t {
        real red[2 * W];
    };

    __device__ static __forceinline__ void run(P const &prm, lds_t &lds, real *scratch_all) {
        const int tid = threadIdx.x;
        real *const red = lds.red;
        real *const scratch = scratch_all + (size_t)blockIdx.x * prm.u_capacity;
        graph_header_t const *const headers = reinterpret_cast<graph_header_t const *>(prm.arena);

        for (unsigned t = blockIdx.x; t < prm.n_launch_jobs; t += gridDim.x) {
            const job_t job = prm.jobs[t];     // jobs of this launch, in launch order
            const Graph g1(prm.arena, headers[job.i]);
            const Graph g2(prm.arena, headers[job.j]);
            const int n1 = g1.n_node, n2 = g2.n_node, N = n1 * n2;
            const int nnz1 = g1.n_nz;
            const long ntask = (long)nnz1 * n2;
            const real q = prm.q, q0 = prm.q0;
            const real inv1q2 = real(1) / ((real(1) - q) * (real(1) - q));
            const real bscale = q * q / (q0 * q0);
            real *const X = scratch;
            real *const Rv = X + (size_t)N * C;
            real *const Pv = Rv + (size_t)N * C;
            real *const U = Pv + (size_t)N * C;

            auto diag = [&](int i, real &dg, real &mi) {
                const int i1 = i / n2, i2 = i - i1 * n2;
                const real dx = real(g1.degree[i1]) * real(g2.degree[i2]) * inv1q2;
                const real vx = prm.node_kernel(g1.node[i1], g2.node[i2]);
                dg = dx / vx;
                mi = vx / dx;
            };

            __syncthreads();   // previous pair is done with the scratch
            real rTz = 0;
            for (int i = tid; i < N; i += TPB) {
                const int i1 = i / n2, i2 = i - i1 * n2;
                real dg, mi;
                diag(i, dg, mi);
                const real dx = real(g1.degree[i1]) * real(g2.degree[i2]) * inv1q2;
                real b[C];
                b[0] = dx * bscale;
                if constexpr (C == 2)
                    b[1] = real(prm.p_start(g1.node[i1])) * real(prm.p_start(g2.node[i2]));
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    X[(size_t)i * C + c] = 0;
                    Rv[(size_t)i * C + c] = b[c];
                    Pv[(size_t)i * C + c] = b[c] * mi;
                    rTz += b[c] * b[c] * mi;
                }
            }
            rTz = block_reduce<real, W>::sum(rTz, red);

            const real tol = (C == 2) ? real(1e-10) * real(2 * N) : prm.ftol * real(N);
            const real tol2 = tol * tol;
            unsigned it = 0;
            for (; it < (unsigned)N && rTz != real(0); ++it) {
                __syncthreads();
                // stage 1
                for (long tk = tid; tk < ntask; tk += TPB) {
                    const int i2 = (int)(tk / nnz1), a = (int)(tk - (long)i2 * nnz1);
                    const nz_t z1 = g1.nz[a];
                    const edge_t e1 = g1.edge[a];
                    real acc[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = 0;
                    for (int b = g2.rowptr[i2]; b < (int)g2.rowptr[i2 + 1]; ++b) {
                        const real e = prm.edge_kernel(e1, g2.edge[b]);
                        const size_t col = (size_t)z1.j * n2 + g2.nz[b].j;
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] += e * Pv[col * C + c];
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) U[(size_t)tk * C + c] = acc[c];
                }
                __syncthreads();
                // stage 2 + p.Ap ; Ap is kept in U's own rows? no: recomputed below
                real pAp = 0;
                for (int i = tid; i < N; i += TPB) {
                    const int i1 = i / n2, i2 = i - i1 * n2;
                    real dg, mi;
                    diag(i, dg, mi);
                    real acc[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = 0;
                    for (int a = g1.rowptr[i1]; a < (int)g1.rowptr[i1 + 1]; ++a)
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] += U[((size_t)i2 * nnz1 + a) * C + c];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const real pv = Pv[(size_t)i * C + c];
                        pAp += pv * (dg * pv - acc[c]);
                    }
                }
                pAp = block_reduce<real, W>::sum(pAp, red);
                if (pAp == real(0)) break;
                const real alpha = rTz / pAp;
                real rTr = 0, rTz_next = 0;
                // x, r update (Ap recomputed from U: U is still valid)
                for (int i = tid; i < N; i += TPB) {
                    const int i1 = i / n2, i2 = i - i1 * n2;
                    real dg, mi;
                    diag(i, dg, mi);
                    real acc[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = 0;
                    for (int a = g1.rowptr[i1]; a < (int)g1.rowptr[i1 + 1]; ++a)
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] += U[((size_t)i2 * nnz1 + a) * C + c];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const size_t k = (size_t)i * C + c;
                        const real pv = Pv[k];
                        const real Ap = dg * pv - acc[c];
                        X[k] += alpha * pv;
                        const real rv = Rv[k] - alpha * Ap;
                        Rv[k] = rv;
                        rTr += rv * rv;
                        rTz_next += rv * rv * mi;
                    }
                }
                block_reduce<real, W>::sum2(rTr, rTz_next, red);
                if (rTr < tol2) {
                    ++it;
                    break;
                }
                const real beta = rTz_next / rTz;
                for (int i = tid; i < N; i += TPB) {
                    real dg, mi;
                    diag(i, dg, mi);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const size_t k = (size_t)i * C + c;
                        Pv[k] = mi * Rv[k] + beta * Pv[k];
                    }
                }
                rTz = rTz_next;
            }
            __syncthreads();
            if (prm.iters != nullptr && tid == 0) prm.iters[prm.order[t]] = it;

            // ---- output (same conventions as pair_solver) ----------------------
            const unsigned flags = prm.flags;
            const unsigned I1 = prm.starts[job.i], I2 = prm.starts[job.j];
            const bool mirror = (flags & F_SYMMETRIC) && job.i != job.j;
            real ksum = 0;
            for (int i = tid; i < N; i += TPB) {
                const int i1 = i / n2, i2 = i - i1 * n2;
                const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                real xi = X[(size_t)i * C];
                if (flags & F_LMIN1) xi -= real(prm.node_kernel(v1, v2)) * bscale;
                const real rv = xi * real(prm.p_start(v1)) * real(prm.p_start(v2));
                ksum += rv;
                if (flags & F_NODAL) {
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
            if (!(flags & F_NODAL)) {
                ksum = block_reduce<real, W>::sum(ksum, red);
                if (tid == 0) {
                    if (flags & F_PACKED) {
                        prm.gramian[prm.order[t]] = ksum;   // slab in job order
                    } else if (flags & F_DIAGONAL) {
                        prm.gramian[I1] = ksum;
                    } else {
                        prm.gramian[(size_t)I1 + (size_t)prm.nX * I2] = ksum;
                        if (mirror) prm.gramian[(size_t)I2 + (size_t)prm.nX * I1] = ksum;
                    }
                }
            }

            if constexpr (C == 2) {
                real jac[n_jac];
#pragma unroll
                for (int j = 0; j < n_jac; ++j) jac[j] = 0;
                const real Q = real(1) / (real(1) - q), Q3 = Q * Q * Q;
                for (int i = tid; i < N; i += TPB) {
                    const int i1 = i / n2, i2 = i - i1 * n2;
                    const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                    const real p1 = prm.p_start(v1), p2 = prm.p_start(v2);
                    const real dox = real(g1.degree[i1]) * real(g2.degree[i2]);
                    const real dx = dox * inv1q2;
                    const real v = prm.node_kernel(v1, v2);
                    const real YDq = X[(size_t)i * 2], Yp = X[(size_t)i * 2 + 1];
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
                    for (long tk = tid; tk < ntask; tk += TPB) {
                        const int i2 = (int)(tk / nnz1), a = (int)(tk - (long)i2 * nnz1);
                        const nz_t z1 = g1.nz[a];
                        const edge_t e1 = g1.edge[a];
                        const real Yp = X[((size_t)z1.i * n2 + i2) * 2 + 1];
                        for (int b = g2.rowptr[i2]; b < (int)g2.rowptr[i2 + 1]; ++b) {
                            auto de = prm.edge_kernel._j_a_c_o_b_i_a_n_(e1, g2.edge[b]);
                            const real w = Yp * X[((size_t)z1.j * n2 + g2.nz[b].j) * 2];
#pragma unroll
                            for (int j = 0; j < EdgeK::jac_dims; ++j) jac[off_e + j] += w * real(de[j]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < n_jac; ++j) {
                    const real g = block_reduce<real, W>::sum(jac[j], red);
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
};

}  // namespace mgk
}  // namespace graphdot
#endif
