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
//    every lane keeps S (value, row, column) triples of off-diagonal nonzeros
//    -- kappa_e(e1, e2) * w1 * w2 is evaluated once, not once per CG iteration
//    (the reference re-evaluates the edge kernel for every nonzero in every
//    iteration) -- plus the R rows of x, r, p and the Jacobi diagonal it owns;
//  * the only per-iteration memory traffic is LDS: p is published to LDS, the
//    mat-vec gathers p[col] with ds_read and scatters with ds_add_f32 into
//    Ap[row]; there are no global-memory CG vectors and no global atomics;
//  * 64 consecutive nonzero pairs form a ta x tb tile of (nonzeros of G1) x
//    (nonzeros of G2); since the packer orders nonzeros so that neighbouring
//    entries have distinct sources, the 64 scatter targets of one instruction
//    are distinct rows (no same-address serialisation);
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
    F_PACKED = 32u,    // graph-level: out[job slot] instead of K(I1, I2)
};

struct job_t {
    std::uint32_t i, j;
};

// Kernel parameter block (one by-value kernel argument; packed on the host
// with the same layout by _backend_hip.py).  Microkernel hyperparameters
// travel here (SGPR-resident kernargs) instead of __constant__ symbols
// (reference: _backend_cuda.py:318-340).
template<class real, class Graph, class NodeK, class EdgeK, class PStart> struct params_t {
    Graph const *graphs;
    job_t const *jobs;
    std::uint32_t const *order;  // job ids for this launch; top 3 bits = log2(tb)
    std::uint32_t const *starts;
    real *gramian;
    real *gradient;
    std::uint32_t *iters;        // optional per-job CG iteration counts
    std::uint32_t n_launch_jobs;
    std::uint32_t nX, nY, nJ;
    std::uint32_t flags;
    std::uint32_t order_offset;  // slot of order[0] in the packed output
    real q, q0, eps, ftol, gtol;
    NodeK node_kernel;
    EdgeK edge_kernel;
    PStart p_start;
};

template<class real, int W> struct block_reduce {
    // Sum over all 64*W threads; `scratch` holds 2*W reals.
    __device__ static __forceinline__ void sum2(real &a, real &b, real *scratch) {
        a = wave::sum(a);
        b = wave::sum(b);
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
        real b = 0;
        sum2(a, b, scratch);
        return a;
    }
};

template<int W> __device__ __forceinline__ void job_sync() {
    if constexpr (W > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
}

template<class real> __device__ __forceinline__ void lds_add(real *p, real v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// S: off-diagonal nonzeros per lane, R: rows per lane, W: waves per pair,
// C: right-hand sides (1 = value, 2 = value + analytic gradient).
template<class real, int S, int R, int W, int C, class Graph, class NodeK, class EdgeK, class PStart>
struct pair_solver {
    using P = params_t<real, Graph, NodeK, EdgeK, PStart>;
    using node_t = typename Graph::node_t;
    using edge_t = typename Graph::edge_t;
    constexpr static int T = 64 * W;   // threads per pair
    constexpr static int NV = R * T;   // vector capacity
    constexpr static int WPB = (W == 1) ? 4 : 1;  // independent pairs per workgroup
    constexpr static int threads = 64 * W * WPB;
    constexpr static int n_jac = PStart::jac_dims + 1 + NodeK::jac_dims + EdgeK::jac_dims;
    constexpr static int off_q = PStart::jac_dims;
    constexpr static int off_v = off_q + 1;
    constexpr static int off_e = off_v + NodeK::jac_dims;

    struct lds_t {
        real p[WPB][NV * C];
        real Ap[WPB][NV * C];
        real red[WPB][2 * W];
    };

    __device__ static __forceinline__ void run(P const &prm, lds_t &lds) {
        const int lane = wave::laneid();
        const int slot = (W == 1) ? (threadIdx.x / 64) : 0;          // pair slot in workgroup
        const int tid = (W == 1) ? lane : (int)threadIdx.x;           // thread within pair
        const int wv = (W == 1) ? 0 : (int)(threadIdx.x / 64);        // wave within pair
        real *const lp = lds.p[slot];
        real *const lAp = lds.Ap[slot];
        real *const red = lds.red[slot];

        const unsigned n_units = gridDim.x * WPB;
        for (unsigned t = blockIdx.x * WPB + slot; t < prm.n_launch_jobs; t += n_units) {
            const unsigned ord = prm.order[t];
            const unsigned job_id = ord & 0x1FFFFFFFu;
            const int sh = ord >> 29;  // log2(tb)
            const job_t job = prm.jobs[job_id];
            const Graph g1 = prm.graphs[job.i];
            const Graph g2 = prm.graphs[job.j];
            const int n1 = g1.n_node, n2 = g2.n_node, N = n1 * n2;
            const real q = prm.q, q0 = prm.q0;
            const real inv1q2 = real(1) / ((real(1) - q) * (real(1) - q));
            const real bscale = q * q / (q0 * q0);

            // ---- rows owned by this thread: Jacobi diagonal, start vectors --
            real dg[R], mi[R], x[C][R], r[C][R], p[C][R];
            real rTz = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int i = k * T + tid;
                const bool ok = i < N;
                const int i1 = ok ? i / n2 : 0, i2 = ok ? i - (i / n2) * n2 : 0;
                const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                const real dx = real(g1.degree[i1]) * real(g2.degree[i2]) * inv1q2;
                const real vx = prm.node_kernel(v1, v2);
                dg[k] = ok ? dx / vx : real(0);
                mi[k] = ok ? vx / dx : real(0);
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
            }

            // ---- off-diagonal nonzeros owned by this thread ------------------
            const int tb = 1 << sh, ta = 64 >> sh;
            const int la = lane >> sh, lb = lane & (tb - 1);
            const int nnz1 = g1.n_nz, nnz2 = g2.n_nz;
            const int ntb = (nnz2 + tb - 1) >> sh;
            const int nta = (nnz1 + ta - 1) / ta;
            const int ntiles = nta * ntb;
            real val[S];
            unsigned rc[S];  // (row << 16) | col
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int tile = s * W + wv;
                const int ti = tile / ntb, tj = tile - ti * ntb;
                const int a = ti * ta + la, b = tj * tb + lb;
                const bool ok = tile < ntiles && a < nnz1 && b < nnz2;
                const int ac = ok ? a : 0, bc = ok ? b : 0;
                const nz_t z1 = g1.nz[ac], z2 = g2.nz[bc];
                const edge_t e1 = g1.edge[ac], e2 = g2.edge[bc];
                const real e = prm.edge_kernel(e1, e2);
                val[s] = ok ? e : real(0);
                rc[s] = ok ? ((unsigned)(z1.i * n2 + z2.i) << 16) | (unsigned)(z1.j * n2 + z2.j) : 0u;
            }

            // ---- publish p ---------------------------------------------------
            job_sync<W>();  // previous pair's readers of lp/lAp are done
#pragma unroll
            for (int k = 0; k < R; ++k)
#pragma unroll
                for (int c = 0; c < C; ++c) lp[(k * T + tid) * C + c] = p[c][k];
            rTz = block_reduce<real, W>::sum(rTz, red);

            const real tol = (C == 2) ? real(1e-10) * real(2 * N) : prm.ftol * real(N);
            unsigned it = 0;
            for (; it < (unsigned)N && rTz != real(0); ++it) {
                // Ap = diag . p   (own rows)
#pragma unroll
                for (int k = 0; k < R; ++k)
#pragma unroll
                    for (int c = 0; c < C; ++c) lAp[(k * T + tid) * C + c] = dg[k] * p[c][k];
                job_sync<W>();
                // Ap -= W . p     (register-resident nonzeros, LDS gather/scatter)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (s * W + wv >= ntiles) break;  // wave-uniform: no work left
                    if (val[s] != real(0)) {          // lanes past the tile edge
                        const unsigned col = rc[s] & 0xFFFFu, row = rc[s] >> 16;
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            lds_add(&lAp[row * C + c], -val[s] * lp[col * C + c]);
                    }
                }
                job_sync<W>();
                real Ap[C][R];
                real pAp = 0;
#pragma unroll
                for (int k = 0; k < R; ++k)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        Ap[c][k] = lAp[(k * T + tid) * C + c];
                        pAp += p[c][k] * Ap[c][k];
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
                if (sqrt(rTr) < tol) {
                    ++it;
                    break;
                }
                const real beta = rTz_next / rTz;
#pragma unroll
                for (int k = 0; k < R; ++k)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        p[c][k] = z[c][k] + beta * p[c][k];
                        lp[(k * T + tid) * C + c] = p[c][k];
                    }
                rTz = rTz_next;
                // the p just published is read after the job_sync that follows
                // the Ap = diag.p stores of the next iteration
            }
            if (prm.iters != nullptr && tid == 0) prm.iters[job_id] = it;

            // ---- output ------------------------------------------------------
            const unsigned flags = prm.flags;
            const unsigned I1 = prm.starts[job.i], I2 = prm.starts[job.j];
            const bool mirror = (flags & F_SYMMETRIC) && job.i != job.j;
            real ksum = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int i = k * T + tid;
                const bool ok = i < N;
                const int i1 = ok ? i / n2 : 0, i2 = ok ? i - (i / n2) * n2 : 0;
                const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                real xi = x[0][k];
                if (flags & F_LMIN1) xi -= real(prm.node_kernel(v1, v2)) * bscale;
                const real pp = real(prm.p_start(v1)) * real(prm.p_start(v2));
                const real rv = ok ? xi * pp : real(0);
                ksum += rv;
                if ((flags & F_NODAL) && ok) {
                    if (flags & F_BLOCK) {
                        prm.gramian[I1 + i1 + i2 * n2] = rv;
                    } else if (flags & F_DIAGONAL) {
                        if (i1 == i2) prm.gramian[I1 + i1] = rv;
                    } else {
                        prm.gramian[(size_t)(I1 + i1) + (size_t)prm.nX * (I2 + i2)] = rv;
                        if (mirror) prm.gramian[(size_t)(I2 + i2) + (size_t)prm.nX * (I1 + i1)] = rv;
                    }
                }
            }
            if (!(flags & F_NODAL)) {
                ksum = block_reduce<real, W>::sum(ksum, red);
                if (tid == 0) {
                    if (flags & F_PACKED) {
                        prm.gramian[prm.order_offset + t] = ksum;
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
                    lp[(k * T + tid) * 2 + 0] = x[0][k];
                    lp[(k * T + tid) * 2 + 1] = x[1][k];
                }
                real jac[n_jac];
#pragma unroll
                for (int j = 0; j < n_jac; ++j) jac[j] = 0;
                const real Q = real(1) / (real(1) - q), Q3 = Q * Q * Q;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int i = k * T + tid;
                    const bool ok = i < N;
                    const int i1 = ok ? i / n2 : 0, i2 = ok ? i - (i / n2) * n2 : 0;
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
                job_sync<W>();
                if constexpr (EdgeK::jac_dims > 0) {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int tile = s * W + wv;
                        const int ti = tile / ntb, tj = tile - ti * ntb;
                        const int a = ti * ta + la, b = tj * tb + lb;
                        const bool ok = tile < ntiles && a < nnz1 && b < nnz2;
                        const int ac = ok ? a : 0, bc = ok ? b : 0;
                        const edge_t e1 = g1.edge[ac], e2 = g2.edge[bc];
                        auto de = prm.edge_kernel._j_a_c_o_b_i_a_n_(e1, e2);
                        const unsigned col = rc[s] & 0xFFFFu, row = rc[s] >> 16;
                        const real w = ok ? lp[row * 2 + 1] * lp[col * 2 + 0] : real(0);
#pragma unroll
                        for (int j = 0; j < EdgeK::jac_dims; ++j) jac[off_e + j] += w * real(de[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < n_jac; ++j) {
                    const real g = block_reduce<real, W>::sum(jac[j], red);
                    if (tid == 0) {
                        if (flags & F_PACKED) {
                            prm.gradient[(size_t)(prm.order_offset + t) * n_jac + j] = g;
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
