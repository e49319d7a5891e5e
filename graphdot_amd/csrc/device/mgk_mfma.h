// Dense-tile solver on the matrix cores (MI355X, gfx950): DENSE graphs of at
// most 32 nodes under an edge microkernel that does not look at the labels
// (`Constant`: E[a, b] = c w1(a) w2(b), one separable term).
//
// The north star allows MFMA "only if a dense-tile formulation of the
// product-graph SpMV proves profitable"; the reference's own dense path is a
// tile loop on the vector units (graphdot/cpp/marginalized_kernel.h:283-328).
// With a separable edge kernel the off-diagonal operator is two dense
// products,  Y = (c W1) P W2^T,  and `scripts/mfma_experiment.py` measured the
// formulation below at 82.3 M pairs/s against 4.19 M of the on-the-fly dense
// product on the vector pipe (256 from_ase-like graphs, 88 % dense; DESIGN.md
// section 2).  This is that kernel behind the product's interfaces: graph
// images from the arena, the generated node microkernel and starting
// probability, every output mode of a value solve.
//
//  * one wave per pair; every CG vector in the ACCUMULATOR LAYOUT of
//    v_mfma_f32_32x32x2_f32: register r of lane l holds element (row
//    kappa(r, l / 32), column l % 32), kappa(r, h) = 8 (r / 4) + 4 h + r % 4.
//    Vector updates are elementwise, so any layout serves -- this one makes
//    the operands of both products free:
//  * step 1, U = P^T M1^T: the A operand of MFMA s is P^T[i][k] = P[k][i] with
//    i = l % 32, k = kappa(s, h): register s of P as it is.  (The contraction
//    index may be visited in any order as long as A and B agree: MFMA s takes
//    k = kappa(s, 0) on the lower and kappa(s, 1) on the upper half-wave.)
//    B = M1^T in that order, read once per pair from a dense copy in LDS.
//  * step 2, Y = U^T M2^T = M1 P M2^T: A = register s of U, B = M2^T.  Y comes
//    out indexed like P.  32 MFMAs per mat-vec, no gathers, no transposes.
//
// System, iteration and stopping rules are the reference's
// (marginalized_kernel.h:394-461).  Float only (the reference's arithmetic;
// the f64 MFMA rate equals the vector rate).
#ifndef GRAPHDOT_HIP_MGK_MFMA_H_
#define GRAPHDOT_HIP_MGK_MFMA_H_
#include "mgk_solver.h"

namespace graphdot {
namespace mgk {

template<class real, class Graph, class NodeK, class EdgeK, class PStart>
struct mfma_solver {
    static_assert(sizeof(real) == 4, "the dense-tile solver is a float solver");
    using P = params_t<real, Graph, NodeK, EdgeK, PStart>;
    using node_t = typename Graph::node_t;
    using edge_t = typename Graph::edge_t;
    typedef float v16f __attribute__((ext_vector_type(16)));
    constexpr static int NMAX = 32;

    struct lds_t {
        float W1[NMAX * NMAX];     // c w1: the edge kernel's constant folded in
        float W2[NMAX * NMAX];
    };

    __device__ static __forceinline__ int kappa(int r, int h) { return 8 * (r / 4) + 4 * h + r % 4; }

    __device__ static __forceinline__ void run(P const &prm, lds_t &lds, char *dyn) {
        const int lane = wave::laneid(), col = lane & 31, half = lane >> 5;
        graph_header_t const *const headers = reinterpret_cast<graph_header_t const *>(prm.arena);
        char *const lG1 = dyn, *const lG2 = dyn + prm.g_capacity;

        for (unsigned t = blockIdx.x; t < prm.n_launch_jobs; t += gridDim.x) {
            const job_t job = prm.jobs[t];
            const graph_header_t h1 = headers[job.i], h2 = headers[job.j];
            const int n1 = h1.n_node, n2 = h2.n_node, N = n1 * n2;
            __builtin_amdgcn_wave_barrier();
            {   // both images and the two dense matrices (zero where there is no edge)
                typedef unsigned v4 __attribute__((ext_vector_type(4)));
                const unsigned w1 = (h1.perm + 2u * n1 - h1.degree + 15u) / 16u;
                const unsigned w2 = (h2.perm + 2u * n2 - h2.degree + 15u) / 16u;
                const v4 *const s1 = reinterpret_cast<const v4 *>(prm.arena + h1.degree);
                const v4 *const s2 = reinterpret_cast<const v4 *>(prm.arena + h2.degree);
                for (unsigned w = lane; w < w1; w += 64) reinterpret_cast<v4 *>(lG1)[w] = s1[w];
                for (unsigned w = lane; w < w2; w += 64) reinterpret_cast<v4 *>(lG2)[w] = s2[w];
                for (int k = lane; k < NMAX * NMAX; k += 64) lds.W1[k] = lds.W2[k] = 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            const Graph g1(lG1 - h1.degree, h1);
            const Graph g2(lG2 - h2.degree, h2);
            if (h1.n_nz > 0 && h2.n_nz > 0) {
                // E[a, b] = ek(e1[a], e2[b]) = c w1(a) w2(b): W1 takes
                // ek(e1[a], e2[0]) = c w1(a) w2(0), W2 the ratio w2(b) / w2(0)
                const edge_t ref2 = g2.edge[0];
                const float w20 = float(edge_weight<edge_t>::get(ref2));
                for (int e = lane; e < h1.n_nz; e += 64) {
                    const nz_t z = g1.nz[e];
                    lds.W1[z.i * NMAX + z.j] = float(prm.edge_kernel(g1.edge[e], ref2));
                }
                for (int e = lane; e < h2.n_nz; e += 64) {
                    const nz_t z = g2.nz[e];
                    lds.W2[z.i * NMAX + z.j] = float(edge_weight<edge_t>::get(g2.edge[e])) / w20;
                }
            }
            __builtin_amdgcn_wave_barrier();

            const real q = prm.q, q0 = prm.q0;
            const real inv1q2 = real(1) / ((real(1) - q) * (real(1) - q));
            const real bscale = q * q / (q0 * q0);
            float b1[16], b2[16], dg[16], mi[16], x[16], r[16], p[16];
            float rTz = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = kappa(s, half);
                // (symmetric adjacency: M^T[k][j] = W[j][k] = W[k][j])
                b1[s] = lds.W1[k * NMAX + col];
                b2[s] = lds.W2[k * NMAX + col];
                // element (row i1 = k, column i2 = col)
                const bool live = k < n1 && col < n2;
                const int i1 = live ? k : 0, i2 = live ? col : 0;
                const float dx = g1.degree[i1] * g2.degree[i2] * inv1q2;
                const float v = float(prm.node_kernel(g1.node[i1], g2.node[i2]));
                dg[s] = live ? dx / v : 1.f;
                mi[s] = live ? v / dx : 1.f;
                const float b = live ? dx * bscale : 0.f;
                x[s] = 0.f;
                r[s] = b;
                p[s] = b * mi[s];
                rTz += b * p[s];
            }
            rTz = wave::sum(rTz);
            const float tol = prm.ftol * (float)N, tol2 = tol * tol;
            unsigned it = 0;
            for (; it < (unsigned)N && rTz != 0.f; ++it) {
                v16f U = {0}, Y = {0};
                // U[i2][j1] = sum_i1 P[i1][i2] M1[j1][i1]
#pragma unroll
                for (int s = 0; s < 16; ++s) U = __builtin_amdgcn_mfma_f32_32x32x2f32(p[s], b1[s], U, 0, 0, 0);
                // Y[j1][j2] = sum_i2 U[i2][j1] M2[j2][i2]
#pragma unroll
                for (int s = 0; s < 16; ++s) Y = __builtin_amdgcn_mfma_f32_32x32x2f32(U[s], b2[s], Y, 0, 0, 0);
                float Ap[16], pAp = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    Ap[s] = dg[s] * p[s] - Y[s];
                    pAp += p[s] * Ap[s];
                }
                pAp = wave::sum(pAp);
                if (pAp == 0.f) break;
                const float alpha = rTz / pAp;
                float rTr = 0.f, rTz_next = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    x[s] += alpha * p[s];
                    r[s] -= alpha * Ap[s];
                    const float z = mi[s] * r[s];
                    rTr += r[s] * r[s];
                    rTz_next += r[s] * z;
                }
                wave::sum2(rTr, rTz_next);
                if (rTr < tol2) {
                    ++it;
                    break;
                }
                const float beta = rTz_next / rTz;
#pragma unroll
                for (int s = 0; s < 16; ++s) p[s] = mi[s] * r[s] + beta * p[s];
                rTz = rTz_next;
            }
            if (prm.iters != nullptr && lane == 0) prm.iters[prm.order[t]] = it;

            // ---- output (conventions of pair_solver / template.cu:100-224) ----
            const unsigned flags = prm.flags;
            const unsigned I1 = prm.starts[job.i], I2 = prm.starts[job.j];
            const bool mirror = (flags & F_SYMMETRIC) && job.i != job.j;
            float ksum = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = kappa(s, half);
                const bool live = k < n1 && col < n2;
                const int i1 = live ? k : 0, i2 = live ? col : 0;
                const node_t v1 = g1.node[i1], v2 = g2.node[i2];
                float xi = x[s];
                if (flags & F_LMIN1) xi -= float(prm.node_kernel(v1, v2)) * bscale;
                const float rv = live ? xi * float(prm.p_start(v1)) * float(prm.p_start(v2)) : 0.f;
                ksum += rv;
                if ((flags & F_NODAL) && live) {
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
                ksum = wave::sum(ksum);
                if (lane == 0) {
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
        }
    }
};

}  // namespace mgk
}  // namespace graphdot
#endif
