// Blocked Cholesky factorisation of a dense symmetric positive definite
// matrix in double precision on gfx950, for the Gaussian process caller of the
// marginalized graph kernel (reference: scipy on the host,
// graphdot/model/gaussian_process/base.py:108-127).
//
// A 1000 x 1000 kernel matrix is 0.33 GFLOP: the library factorisation
// (rocSOLVER through torch.linalg.cholesky) is bound by its column-by-column
// dependency chain, 2.7 ms on MI355X, a fifth of a GPR likelihood step.  Here:
// right-looking, 64-column panels, two launches per panel --
//
//   potrf_panel:  one workgroup per 64-row block below the diagonal block,
//                 L_ik = A_ik L_kk^-T, one 64 x 64 x 64 product with the
//                 inverse of the diagonal factor read from a 64 x 64 workspace;
//   syrk_update:  one workgroup per tile of the trailing matrix,
//                 A_ij -= L_ik L_jk^T; the workgroup of the first tile -- the
//                 next panel's diagonal block -- goes on to factor it (in
//                 registers, with the inverse of the factor as a by-product)
//                 and leaves L in place and L^-1 in the workspace;
//
// and one potrf_diag launch (a single workgroup) for the first diagonal block.
// Exactly ONE workgroup ever reads and writes a diagonal block, and the
// launches that consume its factor come behind it in stream order: nothing
// inside a launch depends on another workgroup.  (Round 2 had every workgroup
// of the panel launch factor the diagonal block redundantly from global memory
// while workgroup 0 wrote the factor back in place: a workgroup dispatched
// late could read L_kk for A_kk.)
//
// Row-major, lower triangle, in place; the strict upper triangle of the
// diagonal blocks is zeroed, tiles above the diagonal are left untouched (the
// caller takes tril).  A matrix that is not positive definite ends with NaN on
// the diagonal of L.  Compiled without fast-math: the square roots and
// divisions are the IEEE ones.
#include <hip/hip_runtime.h>

namespace {

constexpr int B = 64;        // panel width = tile edge
constexpr int LD = B + 1;    // LDS row stride of a tile (odd: no bank conflicts
                             // for column walks)

// load a B x B tile (rows r0.., columns c0..) of the n x n row-major matrix;
// out-of-range entries read 0, or 1 on the diagonal when `identity` is set
__device__ __forceinline__ void load_tile(double (*T)[LD], const double *A, int ld, int n,
                                          int r0, int c0, bool identity) {
    for (int e = threadIdx.x; e < B * B; e += blockDim.x) {
        const int r = e / B, c = e % B;
        const int gr = r0 + r, gc = c0 + c;
        double v = 0;
        if (gr < n && gc < n) v = A[(size_t)gr * ld + gc];
        else if (identity && r == c) v = 1;
        T[r][c] = v;
    }
}

// 1 / sqrt(x) to the last bit or two: hardware estimate + two Newton steps
// (the IEEE sqrt and division it replaces are ~80 dependent instructions on
// the critical path of every column)
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return x > 0.0 ? y : __builtin_nan("");
}

// Factorisation of one 64 x 64 diagonal block held in registers by 256
// threads: thread (ti, tj) of a 16 x 16 grid owns the elements
// (ti + 16 u, tj + 16 v), u, v = 0..3, of the block D and of W, which starts as
// the identity.  TWO columns j, j + 1 per barrier: the owners publish columns
// j, j + 1 of D and rows j, j + 1 of W (unscaled, as they stand before the
// step) to one of two LDS buffers; every thread derives the 2 x 2 pivot factor
//     l_jj = sqrt(D_jj),  l_j+1,j = D_j+1,j / l_jj,
//     l_j+1,j+1 = sqrt(D_j+1,j+1 - l_j+1,j^2)
// itself (two inverse square roots instead of a broadcast and a second
// barrier), scales what it needs and applies the rank-2 updates
//     D[r][c] -= l0_r l0_c + l1_r l1_c   (r, c > j + 1),
//     W[r][:] -= l0_r W'[j][:] + l1_r W'[j+1][:]   (r > j + 1),
// i.e. the eliminations that turn D into L also turn the identity into L^-1.
// (The diagonal blocks are half of the factorisation of a 1000 x 1000 matrix:
// 0.4 us per column of a dependent chain -- inverse square root, LDS round
// trip, barrier.  Two columns per barrier: 0.915 -> 0.872 ms at n = 1000.)
struct factor_lds_t {
    double colD[2][2][B];     // [buffer][column j / j + 1][row]
    double rowW[2][2][B];     // [buffer][row j / j + 1][column]
};

__device__ __forceinline__ void factor_block(double (&d)[4][4], double (&w)[4][4], factor_lds_t &s) {
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    // (the quarter jq = j / 16 of the column is a compile-time index of the
    // register tiles: the loop over it is unrolled, the loop over jr is not)
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
#pragma nounroll
        for (int jr = 0; jr < 16; jr += 2) {
            const int j = 16 * jq + jr, buf = (jr >> 1) & 1;
            // owners publish columns j, j + 1 of D and rows j, j + 1 of W
            if ((tj & ~1) == jr) {
#pragma unroll
                for (int u = 0; u < 4; ++u) s.colD[buf][tj & 1][ti + 16 * u] = d[u][jq];
            }
            if ((ti & ~1) == jr) {
#pragma unroll
                for (int v = 0; v < 4; ++v) s.rowW[buf][ti & 1][tj + 16 * v] = w[jq][v];
            }
            __syncthreads();
            double const *const c0 = s.colD[buf][0], *const c1 = s.colD[buf][1];
            double const *const w0 = s.rowW[buf][0], *const w1 = s.rowW[buf][1];
            const double inv0 = rsqrt_f64(c0[j]);             // (NaN if not positive)
            const double lj1 = c0[j + 1] * inv0;              // l_j+1,j
            const double d11 = c1[j + 1] - lj1 * lj1;
            const double inv1 = rsqrt_f64(d11);
            double l0r[4], l1r[4], l0c[4], l1c[4], wj0[4], wj1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = ti + 16 * u, c = tj + 16 * u;
                const double a0 = c0[r] * inv0, a1 = (c1[r] - a0 * lj1) * inv1;
                const double b0 = c0[c] * inv0, b1 = (c1[c] - b0 * lj1) * inv1;
                // columns j and j + 1 keep the scaled entries; rows / columns
                // up to j + 1 take no update
                if (tj == jr && r >= j) d[u][jq] = a0;
                if (tj == jr + 1 && r >= j + 1) d[u][jq] = a1;
                l0r[u] = r > j + 1 ? a0 : 0.0;
                l1r[u] = r > j + 1 ? a1 : 0.0;
                l0c[u] = c > j + 1 ? b0 : 0.0;
                l1c[u] = c > j + 1 ? b1 : 0.0;
                wj0[u] = w0[c] * inv0;
                wj1[u] = (w1[c] - lj1 * wj0[u]) * inv1;
            }
            if (ti == jr) {
#pragma unroll
                for (int v = 0; v < 4; ++v) w[jq][v] = wj0[v];
            }
            if (ti == jr + 1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) w[jq][v] = wj1[v];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    d[u][v] -= l0r[u] * l0c[v] + l1r[u] * l1c[v];
                    w[u][v] -= l0r[u] * wj0[v] + l1r[u] * wj1[v];
                }
            // (no second barrier: the next pair of columns goes to the other
            // buffer, and a thread is at most one barrier ahead of the slowest)
        }
    }
}

// L (lower triangle, zeros above) of the diagonal block at k0 into A, its
// inverse into the dense 64 x 64 row-major workspace
__device__ __forceinline__ void store_factor(double *A, int ld, int n, int k0, double *Linv,
                                             const double (&d)[4][4], const double (&w)[4][4]) {
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = ti + 16 * u, c = tj + 16 * v;
            if (k0 + r < n && k0 + c < n)
                A[(size_t)(k0 + r) * ld + k0 + c] = c <= r ? d[u][v] : 0.0;
            Linv[r * B + c] = c <= r ? w[u][v] : 0.0;
        }
}

}  // namespace

// The first diagonal block (k0 = 0 in a full factorisation): ONE workgroup of
// 256 threads factors it and leaves L in place and L^-1 in `Linv`.
extern "C" __global__ __launch_bounds__(256)
void potrf_diag_f64(double *A, int ld, int n, int k0, double *Linv) {
    __shared__ factor_lds_t s;
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    double d[4][4], w[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = ti + 16 * u, c = tj + 16 * v;
            const int gr = k0 + r, gc = k0 + c;
            // (beyond the matrix: identity, so that the factor stays defined)
            d[u][v] = (gr < n && gc < n) ? A[(size_t)gr * ld + gc] : (r == c ? 1.0 : 0.0);
            w[u][v] = (r == c) ? 1.0 : 0.0;
        }
    factor_block(d, w, s);
    store_factor(A, ld, n, k0, Linv, d, w);
}

// Panel k0 / 64 below its (already factored) diagonal block: workgroup b takes
// the 64-row block b + 1, L_ik = A_ik L_kk^-T.  grid.x = number of 64-row
// blocks below the diagonal block; 256 threads.  Reads `Linv`, never the
// diagonal block.
extern "C" __global__ __launch_bounds__(256)
void potrf_panel_f64(double *A, int ld, int n, int k0, const double *Linv) {
    __shared__ double Li[B][LD];         // L_kk^-1
    __shared__ double C[B][LD];          // this workgroup's block of the panel
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int r0 = k0 + B * ((int)blockIdx.x + 1);     // first row of this block
    load_tile(C, A, ld, n, r0, k0, false);
    for (int e = tid; e < B * B; e += 256) Li[e / B][e % B] = Linv[e];
    __syncthreads();
    // X[r][c] = sum_{p <= c} C[r][p] Linv[c][p]
    // (thread (ti, tj) computes the outputs (ti + 16 u, tj + 16 v): the 16
    // threads of a row read 16 rows of Linv at the odd stride LD -- distinct
    // banks -- and write 16 consecutive columns)
    double x[4][4] = {};
    for (int p = 0; p < B; ++p) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = C[ti + 16 * u][p];
            b[u] = Li[tj + 16 * u][p];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) x[u][v] += a[u] * b[v];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int gr = r0 + ti + 16 * u, gc = k0 + tj + 16 * v;
            if (gr < n && gc < n) A[(size_t)gr * ld + gc] = x[u][v];
        }
}

// Trailing update after panel k0 / 64: tile t of the lower triangle of the
// remaining blocks, A_ij -= L_ik L_jk^T.  grid.x = m (m + 1) / 2 with m the
// number of row blocks below the panel's diagonal block.  Tile 0 is the next
// panel's diagonal block: its workgroup -- the only one that touches it --
// factors the updated block from its registers and writes L in place and
// L^-1 to `Linv` (which the panel launch of k0 has finished reading: stream
// order).
extern "C" __global__ __launch_bounds__(256)
void syrk_update_f64(double *A, int ld, int n, int k0, double *Linv) {
    __shared__ double Ti[B][LD];
    __shared__ double Tj[B][LD];
    __shared__ factor_lds_t s;
    // linear tile index -> (bi, bj), bj <= bi
    int t = blockIdx.x, bi = 0;
    while (t > bi) {
        t -= bi + 1;
        ++bi;
    }
    const int bj = t;
    const int r0 = k0 + B * (bi + 1), c0 = k0 + B * (bj + 1);
    load_tile(Ti, A, ld, n, r0, k0, false);
    load_tile(Tj, A, ld, n, c0, k0, false);
    __syncthreads();
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    double x[4][4] = {};
    for (int p = 0; p < B; ++p) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = Ti[ti + 16 * u][p];
            b[u] = Tj[tj + 16 * u][p];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) x[u][v] += a[u] * b[v];
    }
    if (blockIdx.x == 0) {      // (workgroup-uniform)
        double w[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = ti + 16 * u, c = tj + 16 * v;
                const int gr = r0 + r, gc = c0 + c;
                x[u][v] = (gr < n && gc < n) ? A[(size_t)gr * ld + gc] - x[u][v] : (r == c ? 1.0 : 0.0);
                w[u][v] = (r == c) ? 1.0 : 0.0;
            }
        factor_block(x, w, s);
        store_factor(A, ld, n, r0, Linv, x, w);
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int gr = r0 + ti + 16 * u, gc = c0 + tj + 16 * v;
            if (gr < n && gc < n) A[(size_t)gr * ld + gc] -= x[u][v];
        }
}
