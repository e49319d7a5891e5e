// Cholesky factorisation AND inverse of a dense symmetric positive definite
// matrix in double precision on gfx950, in ONE launch, as a data flow over
// 64 x 64 tiles -- for the Gaussian process caller of the marginalized graph
// kernel (reference: scipy on the host,
// graphdot/model/gaussian_process/base.py:108-127, gpr.py:286-297).
//
// A 1000 x 1000 kernel matrix is 1 GFLOP for factor + inverse: nothing.  What
// it costs is its dependency chain.  The library (rocSOLVER through
// torch.linalg.cholesky) walks it column by column, 2.7 ms; rounds 2-5 had a
// right-looking factorisation of 64-column panels with two launches per panel
// (0.87 ms: every launch drains before the next starts, 55 us per panel)
// followed by two library calls for L^-1 and K^-1 = L^-T L^-1 -- 1.24 ms in
// all, replicated on every rank, half of an 8-rank likelihood step (DESIGN
// section 8).  Here every tile of three block matrices is a ROLE that one
// workgroup carries from its first term to its last, in registers:
//
//   A(i, j), j <= i:  S = sum_{k<j} L_ik L_jk^T as the tiles L_.k appear;
//                     then i == j: factor A_jj - S in registers (factor_block:
//                     L_jj and L_jj^-1 together), else L_ij = (A_ij - S) L_jj^-T;
//   Z(i, j), i <  j:  the blocks of Z = L^-T (upper triangular; Z L^T = I):
//                     Z_ij = -(sum_{k=i}^{j-1} Z_ik L_jk^T) L_jj^-T, Z_ii = L_ii^-T;
//   Kinv(i, j), j <= i:  (K^-1)_ij = sum_{k>=i} Z_ik Z_jk^T, written with its
//                     mirror image.
//
// All three are the same step, S += P Q^T on two row-major tiles staged in
// LDS, on the double-precision matrix cores (v_mfma_f64_16x16x4_f64: one
// operand register per lane and k-step where the vector form re-reads eight
// LDS values per sixteen FMAs; same peak).  A role waits for a tile on a flag
// word in device memory; the producer has stored the tile with sc1 stores
// (write-through, they leave the XCD's L2), every wave has drained its stores,
// one lane then stores the flag; the consumer polls the flag with an sc1 load
// and reads the tile with sc1 loads (they bypass the L1): no cache-wide
// write-back or invalidate on the way (1.7-6.5 us each on this chip, two per
// panel on the critical path; measured here: 0.6 us from the producer's flag
// store to the consumer's exit from its poll).  No address is read before its
// final value is written, except by its own producer.
//
// Roles are numbered in an order in which every role depends on earlier ones
// only (column by column: A(j,j), A(j+1..,j), Z(0..j-1,j); then Kinv) and are
// handed out by an atomic counter, so a role is only ever held by a RUNNING
// workgroup and so are all roles it waits for: no deadlock whatever part of
// the grid is resident, no cooperative launch.  A wait that lasts ~2 s
// poisons the launch (status word; the host raises).
//
// The critical path is FACTOR(j) -> L_j+1,j -> update of A_j+1,j+1 ->
// FACTOR(j+1); scripts/potrf_timeline.py prints it from in-kernel stamps.
//
// Row-major, lower triangle, in place; the strict upper triangle of the
// diagonal blocks is zeroed, tiles above the diagonal are left untouched (the
// caller takes tril).  A matrix that is not positive definite ends with NaN on
// the diagonal of L and in the log-determinant.  Compiled without fast-math:
// the arithmetic is IEEE.
#include <hip/hip_runtime.h>

namespace {

constexpr int B = 64;        // tile edge
constexpr int LT = B + 2;    // LDS row stride of a staged tile: the 16 rows x 2
                             // k-columns a half-wave reads as MFMA operands
                             // fall on 32 distinct bank pairs (2 row + k mod 32)

typedef double v4d __attribute__((ext_vector_type(4)));

// 1 / sqrt(x) to the last bit or two: the hardware estimate (good to ~2^-23)
// and ONE third-order step, y (1 + e / 2 + 3 e^2 / 8) with e = 1 - x y^2 --
// five dependent operations where two Newton steps are eight (the IEEE sqrt
// and division it replaces are ~80, on the critical path of every column)
__device__ __forceinline__ double rsqrt_f64(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    const double q = e * __builtin_fma(e, 0.375, 0.5);
    return x > 0.0 ? __builtin_fma(y, q, y) : __builtin_nan("");
}

// Factorisation of one 64 x 64 diagonal block held in registers by 256
// threads: thread (ti, tj) of a 16 x 16 grid owns the elements
// (ti + 16 u, tj + 16 v), u, v = 0..3, of the block D (its lower triangle is
// what counts) and of W, which starts as the identity.  TWO columns j, j + 1
// per barrier: the owners publish columns j, j + 1 of D and rows j, j + 1 of W
// (unscaled, as they stand before the step) to one of two LDS buffers; every
// thread derives the 2 x 2 pivot factor
//     l_jj = sqrt(D_jj),  l_j+1,j = D_j+1,j / l_jj,
//     l_j+1,j+1 = sqrt(D_j+1,j+1 - l_j+1,j^2)
// itself (two inverse square roots instead of a broadcast and a second
// barrier), scales what it needs and applies the rank-2 updates
//     D[r][c] -= l0_r l0_c + l1_r l1_c   (r >= c > j + 1),
//     W[r][c] -= l0_r W'[j][c] + l1_r W'[j+1][c]   (r > j + 1, c <= j + 1),
// i.e. the eliminations that turn D into L also turn the identity into L^-1.
// Round 6: the quarter jq = j / 16 of the step is a compile-time fact (one
// instantiation per quarter), and with it WHICH of the thread's sixteen 1 x 1
// blocks a step can touch: D only where jq <= v <= u (rows and columns of
// finished quarters are final, the strict upper triangle is never read), W
// only where u >= jq >= v (W stays lower triangular) -- 20 + 20 block updates
// over the four quarters where the first form did 64 + 64, each one two FMAs
// (the compiler had made multiply, multiply-add and subtract of them).
struct factor_lds_t {
    double colD[2][2][B];     // [buffer][column j / j + 1][row]
    double rowW[2][2][B];     // [buffer][row j / j + 1][column]
};

template<int JQ>
__device__ __forceinline__ void factor_quarter(double (&d)[4][4], double (&w)[4][4], factor_lds_t &s) {
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    // the factors of the step before, for the blocks whose update waits
    double pa0[4], pa1[4], pb0[4], pb1[4], pw0[4], pw1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pa0[u] = pa1[u] = pb0[u] = pb1[u] = pw0[u] = pw1[u] = 0.0;
#pragma nounroll
    for (int jr = 0; jr < 16; jr += 2) {
        const int j = 16 * JQ + jr, buf = (jr >> 1) & 1;
        // owners publish columns j, j + 1 of D (rows of this quarter and
        // below) and rows j, j + 1 of W (columns up to this quarter)
        if ((tj & ~1) == jr) {
#pragma unroll
            for (int u = JQ; u < 4; ++u) s.colD[buf][tj & 1][ti + 16 * u] = d[u][JQ];
        }
        if ((ti & ~1) == jr) {
#pragma unroll
            for (int v = 0; v <= JQ; ++v) s.rowW[buf][ti & 1][tj + 16 * v] = w[JQ][v];
        }
        __syncthreads();
        double const *const c0 = s.colD[buf][0], *const c1 = s.colD[buf][1];
        double const *const w0 = s.rowW[buf][0], *const w1 = s.rowW[buf][1];
        const double inv0 = rsqrt_f64(c0[j]);             // (NaN if not positive)
        const double lj1 = c0[j + 1] * inv0;              // l_j+1,j
        const double d11 = __builtin_fma(-lj1, lj1, c1[j + 1]);
        const double inv1 = rsqrt_f64(d11);
        // -- the blocks OUTSIDE block column JQ of D and block row JQ of W
        // take the update of the step BEFORE, now: nothing of them is
        // published in this quarter, and their FMAs fill the latencies of
        // the chain above (LDS read, two inverse square roots) instead of
        // standing behind it
#pragma unroll
        for (int u = JQ + 1; u < 4; ++u) {
#pragma unroll
            for (int v = JQ + 1; v <= u; ++v)
                d[u][v] = __builtin_fma(-pa0[u], pb0[v], __builtin_fma(-pa1[u], pb1[v], d[u][v]));
#pragma unroll
            for (int v = 0; v <= JQ; ++v)
                w[u][v] = __builtin_fma(-pa0[u], pw0[v], __builtin_fma(-pa1[u], pw1[v], w[u][v]));
        }
        // -- this step's factors
#pragma unroll
        for (int u = JQ; u < 4; ++u) {
            const int r = ti + 16 * u;
            pa0[u] = c0[r] * inv0;
            pa1[u] = __builtin_fma(-pa0[u], lj1, c1[r]) * inv1;
        }
#pragma unroll
        for (int v = JQ; v < 4; ++v) {
            const int c = tj + 16 * v;
            pb0[v] = c0[c] * inv0;
            pb1[v] = __builtin_fma(-pb0[v], lj1, c1[c]) * inv1;
        }
#pragma unroll
        for (int v = 0; v <= JQ; ++v) {
            const int c = tj + 16 * v;
            pw0[v] = w0[c] * inv0;
            pw1[v] = __builtin_fma(-lj1, pw0[v], w1[c]) * inv1;
        }
        // -- block column JQ of D and block row JQ of W at once: the next
        // step publishes from them.  (Rows and columns up to j + 1 take no
        // update: they exist in quarter JQ only.)
        {
            const int rq = ti + 16 * JQ, cq = tj + 16 * JQ;
            const double m0 = cq > j + 1 ? pb0[JQ] : 0.0, m1 = cq > j + 1 ? pb1[JQ] : 0.0;
#pragma unroll
            for (int u = JQ; u < 4; ++u) {
                const int r = ti + 16 * u;
                const double a0 = (u > JQ || r > j + 1) ? pa0[u] : 0.0;
                const double a1 = (u > JQ || r > j + 1) ? pa1[u] : 0.0;
                double x = __builtin_fma(-a0, m0, __builtin_fma(-a1, m1, d[u][JQ]));
                // columns j and j + 1 keep the scaled entries
                if (tj == jr && r >= j) x = pa0[u];
                if (tj == jr + 1 && r >= j + 1) x = pa1[u];
                d[u][JQ] = x;
            }
            const double a0 = rq > j + 1 ? pa0[JQ] : 0.0, a1 = rq > j + 1 ? pa1[JQ] : 0.0;
#pragma unroll
            for (int v = 0; v <= JQ; ++v) {
                double x = __builtin_fma(-a0, pw0[v], __builtin_fma(-a1, pw1[v], w[JQ][v]));
                if (ti == jr) x = pw0[v];
                if (ti == jr + 1) x = pw1[v];
                w[JQ][v] = x;
            }
        }
        // (no second barrier: the next pair of columns goes to the other
        // buffer, and a thread is at most one barrier ahead of the slowest)
    }
    // the last step's update of the waiting blocks
#pragma unroll
    for (int u = JQ + 1; u < 4; ++u) {
#pragma unroll
        for (int v = JQ + 1; v <= u; ++v)
            d[u][v] = __builtin_fma(-pa0[u], pb0[v], __builtin_fma(-pa1[u], pb1[v], d[u][v]));
#pragma unroll
        for (int v = 0; v <= JQ; ++v)
            w[u][v] = __builtin_fma(-pa0[u], pw0[v], __builtin_fma(-pa1[u], pw1[v], w[u][v]));
    }
}

__device__ __forceinline__ void factor_block(double (&d)[4][4], double (&w)[4][4], factor_lds_t &s) {
    factor_quarter<0>(d, w, s);
    factor_quarter<1>(d, w, s);
    factor_quarter<2>(d, w, s);
    factor_quarter<3>(d, w, s);
}

// device-scope accesses of everything one workgroup hands to another
__device__ __forceinline__ double ld_sc1(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a tile as a role reads it: `rows` x `cols` numbers at `p` (row stride `ld`),
// zero beyond; `transposed`: the staged tile is the transpose of what lies
// there (Z_ii = L_ii^-T from the L_ii^-1 the factoring workgroup stored)
struct tile_src_t {
    const double *p;
    int ld, rows, cols;
    bool transposed;
};

__device__ __forceinline__ void stage_tile(double (*T)[LT], const tile_src_t &s) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = threadIdx.x + 256 * q, r = e >> 6, c = e & 63;
        v[q] = (r < s.rows && c < s.cols) ? ld_sc1(s.p + (size_t)r * s.ld + c) : 0.0;
    }
    if (s.transposed) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = threadIdx.x + 256 * q;
            T[e & 63][e >> 6] = v[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = threadIdx.x + 256 * q;
            T[e >> 6][e & 63] = v[q];
        }
    }
}

// The accumulator of a role: wave w of the four holds the 32 x 32 quadrant
// (w >> 1, w & 1) of the tile as 2 x 2 MFMA blocks of 16 x 16; element `g` of
// block (rb, cb) in lane l is row 16 rb + (l >> 4) + 4 g, column 16 cb + (l & 15)
// of the quadrant (the f64 C/D map of the instruction).
struct acc_t {
    v4d c[2][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) c[rb][cb] = v4d{0.0, 0.0, 0.0, 0.0};
    }
    __device__ __forceinline__ static int row(int rb, int g) {
        return 32 * ((threadIdx.x >> 6) >> 1) + 16 * rb + ((threadIdx.x & 63) >> 4) + 4 * g;
    }
    __device__ __forceinline__ static int col(int cb) {
        return 32 * ((threadIdx.x >> 6) & 1) + 16 * cb + (threadIdx.x & 15);
    }
    // S += P Q^T (both tiles staged, 64 x 64, row-major)
    __device__ __forceinline__ void mma(const double (*P)[LT], const double (*Q)[LT]) {
        const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
        const double *const p0 = &P[32 * (w >> 1) + (l & 15)][l >> 4];
        const double *const q0 = &Q[32 * (w & 1) + (l & 15)][l >> 4];
#pragma unroll 4
        for (int s = 0; s < B; s += 4) {
            const double a0 = p0[s], a1 = p0[16 * LT + s];
            const double b0 = q0[s], b1 = q0[16 * LT + s];
            c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c[0][0], 0, 0, 0);
            c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c[0][1], 0, 0, 0);
            c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c[1][0], 0, 0, 0);
            c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c[1][1], 0, 0, 0);
        }
    }
    // the tile as a staged operand
    __device__ __forceinline__ void to_lds(double (*T)[LT], double sign) const {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) T[row(rb, g)][col(cb)] = sign * c[rb][cb][g];
    }
};

struct sync_t {
    unsigned *next_role, *status, *flag_l, *flag_z;
    int *lds_word;
    // wait until both flags are set (either may be null); false: poisoned
    __device__ __forceinline__ bool wait(const unsigned *f0, const unsigned *f1) const {
        if (threadIdx.x == 0) {
            int ok = 1;
            unsigned spins = 0;
            for (int h = 0; h < 2 && ok; ++h) {
                const unsigned *const f = h ? f1 : f0;
                if (f == nullptr) continue;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 0xFFu) == 0u &&
                        (spins > (1u << 21) ||
                         __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
            }
            *lds_word = ok;
        }
        __syncthreads();
        const int ok = *lds_word;
        __syncthreads();
        return ok != 0;
    }
    // every wave has stored its part of a tile with sc1 stores: drain them,
    // meet, one lane raises the flag
    __device__ __forceinline__ void publish(unsigned *f) const {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

struct spd_args_t {
    double *A;          // n x n row-major, symmetric on entry; L in its lower tiles on exit
    double *Kinv;       // n x n row-major (row stride ldk): the inverse, both triangles
    double *Z;          // nb x nb tiles of 64 x 64: the blocks of L^-T above the diagonal
    double *Linv;       // nb tiles of 64 x 64: L_jj^-1
    unsigned *sync;     // zeroed: [next role, status, pad x 14]
                        //         [nb doubles: sum of log L_rr over the rows of block j]
                        //         [flag_l nb x nb][flag_z nb x nb]
    int ld, ldk, n, invert;
    unsigned long long *stamps;   // null, or 16 wall-clock stamps (100 MHz) per role: scripts/potrf_timeline.py
};

}  // namespace

extern "C" __global__ __launch_bounds__(256)
void spd_factor_invert_f64(spd_args_t a) {
    __shared__ double P[B][LT];
    __shared__ double Q[B][LT];
    __shared__ factor_lds_t fs;
    __shared__ int word;
    __shared__ double red[16];
    const int n = a.n, nb = (n + B - 1) / B;
    const int tid = threadIdx.x;
    sync_t sy;
    sy.next_role = a.sync;
    sy.status = a.sync + 1;
    double *const logdet = reinterpret_cast<double *>(a.sync + 16);
    sy.flag_l = a.sync + 16 + 2 * nb;
    sy.flag_z = sy.flag_l + nb * nb;
    sy.lds_word = &word;
    const int n_a = nb * (nb + 1) / 2;
    const int total = nb * nb + (a.invert ? n_a : 0);
    auto rows_of = [&](int i) { return min(B, n - B * i); };
    auto l_tile = [&](int i, int k) {      // L_ik (k < i) where A holds it
        return tile_src_t{a.A + (size_t)(B * i) * a.ld + B * k, a.ld, rows_of(i), B, false};
    };
    auto z_tile = [&](int i, int k) {      // Z_ik, i <= k
        if (i == k) return tile_src_t{a.Linv + ((size_t)i << 12), B, B, B, true};
        return tile_src_t{a.Z + ((size_t)(i * nb + k) << 12), B, B, B, false};
    };
    auto z_flag = [&](int i, int k) { return i == k ? sy.flag_l + i * nb + i : sy.flag_z + i * nb + k; };
    int role = 0;
    auto stamp = [&](int slot) {
        if (a.stamps != nullptr && tid == 0) a.stamps[(size_t)role * 16 + slot] = wall_clock64();
    };

    for (;;) {
        if (tid == 0) word = (int)atomicAdd(sy.next_role, 1u);
        __syncthreads();
        role = word;
        __syncthreads();
        if (role >= total) return;
        stamp(0);
        acc_t S;
        S.zero();
        if (role < nb * nb) {
            const int j = role / nb, p = role % nb;
            const bool is_a = p < nb - j;
            const int i = is_a ? j + p : p - (nb - j);
            if (!is_a && !a.invert) continue;
            if (is_a && i == j) {
                // -- the diagonal block.  Its own entries are fetched before
                // the wait for the tiles to its left (they are the caller's:
                // nothing in this launch writes them before this role does)
                const int ti = tid >> 4, tj = tid & 15;
                double d[4][4], w[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = ti + 16 * u, c = tj + 16 * v;
                        const int gr = B * j + r, gc = B * j + c;
                        d[u][v] = (gr < n && gc < n) ? ld_sc1(a.A + (size_t)gr * a.ld + gc)
                                                     : (r == c ? 1.0 : 0.0);
                        w[u][v] = (r == c) ? 1.0 : 0.0;
                    }
                for (int k = 0; k < j; ++k) {
                    if (!sy.wait(sy.flag_l + j * nb + k, nullptr)) return;
                    stamp(1);
                    stage_tile(P, l_tile(j, k));
                    __syncthreads();
                    stamp(2);
                    S.mma(P, P);
                    __syncthreads();
                    stamp(3);
                }
                // A_jj - S in the layout of the factorisation
                S.to_lds(P, 1.0);
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (B * j + max(ti + 16 * u, tj + 16 * v) < n) d[u][v] -= P[ti + 16 * u][tj + 16 * v];
                stamp(4);
                factor_block(d, w, fs);
                stamp(5);
                // what the roles below and to the right wait for is L_jj^-1
                double *const li = a.Linv + ((size_t)j << 12);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = ti + 16 * u, c = tj + 16 * v;
                        st_sc1(li + r * B + c, c <= r ? w[u][v] : 0.0);
                    }
                sy.publish(sy.flag_l + j * nb + j);
                stamp(6);
                // (off the critical path: L_jj for the caller and the block's
                // share of log det L)
                double lg = 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = ti + 16 * u, c = tj + 16 * v;
                        const int gr = B * j + r, gc = B * j + c;
                        if (gr < n && gc < n) st_sc1(a.A + (size_t)gr * a.ld + gc, c <= r ? d[u][v] : 0.0);
                        if (r == c) lg += log(d[u][v]);
                    }
                if (ti == tj) red[ti] = lg;
                __syncthreads();
                if (tid == 0) {
                    double t = 0.0;
                    for (int q = 0; q < 16; ++q) t += red[q];
                    logdet[j] = t;
                }
                __syncthreads();
                continue;
            }
            // -- the sum over the earlier columns
            for (int k = is_a ? 0 : i; k < j; ++k) {
                if (!sy.wait(is_a ? sy.flag_l + i * nb + k : z_flag(i, k), sy.flag_l + j * nb + k)) return;
                stamp(1);
                stage_tile(P, is_a ? l_tile(i, k) : z_tile(i, k));
                stage_tile(Q, l_tile(j, k));
                __syncthreads();
                stamp(2);
                S.mma(P, Q);
                __syncthreads();
                stamp(3);
            }
            // -- (A_ij - S) L_jj^-T, or -S L_jj^-T for a block of Z
            if (is_a) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int gr = B * i + acc_t::row(rb, g), gc = B * j + acc_t::col(cb);
                            const double x = gr < n ? ld_sc1(a.A + (size_t)gr * a.ld + gc) : 0.0;
                            S.c[rb][cb][g] = x - S.c[rb][cb][g];
                        }
                S.to_lds(P, 1.0);
            } else {
                S.to_lds(P, -1.0);
            }
            stamp(4);
            if (!sy.wait(sy.flag_l + j * nb + j, nullptr)) return;
            stamp(5);
            stage_tile(Q, tile_src_t{a.Linv + ((size_t)j << 12), B, B, B, false});
            __syncthreads();
            stamp(6);
            S.zero();
            S.mma(P, Q);
            stamp(7);
            double *const out = is_a ? a.A + (size_t)(B * i) * a.ld + B * j
                                     : a.Z + ((size_t)(i * nb + j) << 12);
            const int old = is_a ? a.ld : B, lim = is_a ? rows_of(i) : B;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int r = acc_t::row(rb, g), c = acc_t::col(cb);
                        if (r < lim) st_sc1(out + (size_t)r * old + c, S.c[rb][cb][g]);
                    }
            sy.publish((is_a ? sy.flag_l : sy.flag_z) + i * nb + j);
            stamp(8);
            __syncthreads();
            continue;
        }
        // -- a tile of the inverse: sum_{k >= i} Z_ik Z_jk^T, j <= i
        int t = role - nb * nb, i = 0;
        while (t > i) {
            t -= i + 1;
            ++i;
        }
        const int j = t;
        for (int k = i; k < nb; ++k) {
            if (!sy.wait(z_flag(i, k), i == j ? nullptr : z_flag(j, k))) return;
            stage_tile(P, z_tile(i, k));
            if (i != j) stage_tile(Q, z_tile(j, k));
            __syncthreads();
            S.mma(P, i == j ? P : Q);
            __syncthreads();
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int gr = B * i + acc_t::row(rb, g), gc = B * j + acc_t::col(cb);
                    if (gr < n && gc < n) {
                        a.Kinv[(size_t)gr * a.ldk + gc] = S.c[rb][cb][g];
                        if (i != j) a.Kinv[(size_t)gc * a.ldk + gr] = S.c[rb][cb][g];
                    }
                }
    }
}
