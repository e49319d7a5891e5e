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

// (hooks of scripts/micro/factor_probe.hip, which times the factorisation of
// one block in isolation with parts of a step taken out; the product compiles
// with the defaults)
#ifndef GD_POTRF_STEP_BARRIER
#define GD_POTRF_STEP_BARRIER() __syncthreads()
#endif
#ifndef GD_POTRF_RCP
#define GD_POTRF_RCP(x) rcp_f64(x)
#endif
#ifndef GD_POTRF_STEP_LOOP
#define GD_POTRF_STEP_LOOP _Pragma("nounroll")
#endif

namespace {

constexpr int B = 64;        // tile edge
constexpr int LT = B + 2;    // LDS row stride of a staged tile: the 16 rows x 2
                             // k-columns a half-wave reads as MFMA operands
                             // fall on 32 distinct bank pairs (2 row + k mod 32)

typedef double v4d __attribute__((ext_vector_type(4)));

// 1 / sqrt(x) and 1 / x to the last bit or two: the hardware estimates (good
// to ~2^-23) and ONE third-order step each -- y (1 + e / 2 + 3 e^2 / 8) with
// e = 1 - x y^2; r (1 + e + e^2) with e = 1 - x r -- three to five dependent
// operations where the IEEE sqrt and division are ~80
__device__ __forceinline__ double rsqrt_f64(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    const double q = e * __builtin_fma(e, 0.375, 0.5);
    return x > 0.0 ? __builtin_fma(y, q, y) : __builtin_nan("");
}
__device__ __forceinline__ double rcp_f64(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// Factorisation of one 64 x 64 diagonal block held in registers by 256
// threads: thread (ti, tj) of a 16 x 16 grid owns the elements
// (ti + 16 u, tj + 16 v), u, v = 0..3, of the block D (its lower triangle is
// what counts) and of W, which starts as the identity; the eliminations that
// turn D into its factor turn W into the factor's inverse.
//
// What binds it (scripts/micro/f64_latency.hip, scripts/potrf_timeline.py):
// one wave per SIMD ISSUES a double-precision instruction every 2.4 ns,
// dependent or not, and the publish -> barrier -> read round trip through LDS
// is 60 ns; the first form of the round -- two columns per barrier, every
// thread scaling and masking all sixteen of its blocks in every step, ~270
// instructions -- took 0.75 us per step, 24 us per block, 57 % of a panel's
// time.  So the step is written for its instruction count:
//
//  * root-free, D = L' P L'^T with unit lower L': a step needs the two pivots'
//    reciprocals (one estimate + three operations each) where the Cholesky
//    step needs two inverse square roots one after the other, and only the ROW
//    factors are scaled -- l'_rj = x_rj / p_j; the column side of the update
//    takes the raw column (D[r][c] -= l'_rj x_cj: x_cj = l'_cj p_j).  The
//    square roots come once per block, at the end, off the chain:
//    L = L' sqrt(P), L^-1 = P^-1/2 L'^-1;
//  * the quarter jq = j / 16 of a step is a compile-time fact (one
//    instantiation per quarter) and with it which of the thread's sixteen 1 x 1
//    blocks a step can touch: D only where jq <= v <= u, W only where
//    u >= jq >= v -- 20 + 20 block updates over the four quarters instead of
//    64 + 64, each two FMAs;
//  * nothing is masked: finished columns of L' and finished rows of L'^-1 are
//    CAPTURED in two LDS tiles the moment their owners have them (the staging
//    buffers of the role are free meanwhile), so the register copies of
//    finished rows and columns, and the strict upper triangle, may take any
//    update -- no compare, no select in the step (the first form: ~40 of
//    them).  Valid entries only ever read valid factors;
//  * the blocks outside block column jq of D and block row jq of W take a
//    step's update one step LATE, among the next step's chain (pivot
//    reciprocals): nothing of them is published within the quarter.
//
// A pivot that is not positive goes through the elimination as it is and
// makes its column of L NaN at the end (the inverse square root refuses it).
struct factor_lds_t {
    double colD[2][2][B];     // [buffer][column j / j + 1][row]
    double rowW[2][2][B];     // [buffer][row j / j + 1][column]
    double piv[B], scal[B], rscal[B];     // p_c, sqrt(p_c), 1 / sqrt(p_c)
};

template<int JQ>
__device__ __forceinline__ void factor_quarter(double (&d)[4][4], double (&w)[4][4], factor_lds_t &s,
                                               double (*Lt)[LT], double (*Wt)[LT]) {
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    // the factors of the step before, for the blocks whose update waits
    double pa0[4], pa1[4], pb0[4], pb1[4], pw0[4], pw1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pa0[u] = pa1[u] = pb0[u] = pb1[u] = pw0[u] = pw1[u] = 0.0;
    GD_POTRF_STEP_LOOP
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
        GD_POTRF_STEP_BARRIER();
        double const *const c0 = s.colD[buf][0], *const c1 = s.colD[buf][1];
        double const *const w0 = s.rowW[buf][0], *const w1 = s.rowW[buf][1];
        // the 2 x 2 pivot: p_j, l'_j+1,j, p_j+1 (every thread for itself)
        const double p0 = c0[j], x01 = c0[j + 1];
        const double r0 = GD_POTRF_RCP(p0);
        const double m = x01 * r0;
        const double p1 = __builtin_fma(-m, x01, c1[j + 1]);
        const double r1 = GD_POTRF_RCP(p1);
        if (tid == 0) {
            s.piv[j] = p0;
            s.piv[j + 1] = p1;
        }
        // -- the waiting blocks take the update of the step before
#pragma unroll
        for (int u = JQ + 1; u < 4; ++u) {
#pragma unroll
            for (int v = JQ + 1; v <= u; ++v)
                d[u][v] = __builtin_fma(-pa0[u], pb0[v], __builtin_fma(-pa1[u], pb1[v], d[u][v]));
#pragma unroll
            for (int v = 0; v <= JQ; ++v)
                w[u][v] = __builtin_fma(-pa0[u], pw0[v], __builtin_fma(-pa1[u], pw1[v], w[u][v]));
        }
        // -- this step's factors: rows l'_rj, l'_r,j+1; columns x_cj and
        // x_c,j+1 less its share of column j; rows j, j + 1 of L'^-1
#pragma unroll
        for (int u = JQ; u < 4; ++u) {
            const int r = ti + 16 * u;
            const double x0 = c0[r];
            pa0[u] = x0 * r0;
            pa1[u] = __builtin_fma(-x0, m, c1[r]) * r1;
        }
#pragma unroll
        for (int v = JQ; v < 4; ++v) {
            const int c = tj + 16 * v;
            pb0[v] = c0[c];
            pb1[v] = __builtin_fma(-pb0[v], m, c1[c]);
        }
#pragma unroll
        for (int v = 0; v <= JQ; ++v) {
            const int c = tj + 16 * v;
            pw0[v] = w0[c];
            pw1[v] = __builtin_fma(-m, pw0[v], w1[c]);
        }
        // -- finished: columns j, j + 1 of L' and rows j, j + 1 of L'^-1
        if ((tj & ~1) == jr) {
#pragma unroll
            for (int u = JQ; u < 4; ++u) Lt[ti + 16 * u][j + (tj & 1)] = (tj & 1) ? pa1[u] : pa0[u];
        }
        if ((ti & ~1) == jr) {
#pragma unroll
            for (int v = 0; v <= JQ; ++v) Wt[j + (ti & 1)][tj + 16 * v] = (ti & 1) ? pw1[v] : pw0[v];
        }
        // -- block column JQ of D and block row JQ of W at once: the next
        // step publishes from them
#pragma unroll
        for (int u = JQ; u < 4; ++u)
            d[u][JQ] = __builtin_fma(-pa0[u], pb0[JQ], __builtin_fma(-pa1[u], pb1[JQ], d[u][JQ]));
#pragma unroll
        for (int v = 0; v <= JQ; ++v)
            w[JQ][v] = __builtin_fma(-pa0[JQ], pw0[v], __builtin_fma(-pa1[JQ], pw1[v], w[JQ][v]));
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

// D (registers; A_jj - S on entry) -> the tiles Lt = L' and Wt = L'^-1 in LDS
// (rows x columns, strict upper triangles undefined) and the pivots in `s`:
// L[r][c] = Lt[r][c] s.scal[c] below the diagonal, s.scal[c] on it;
// L^-1[r][c] = Wt[r][c] s.rscal[r] on and below it.
__device__ __forceinline__ void factor_block(double (&d)[4][4], factor_lds_t &s, double (*Lt)[LT],
                                             double (*Wt)[LT]) {
    double w[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) w[u][v] = ((threadIdx.x >> 4) == (threadIdx.x & 15) && u == v) ? 1.0 : 0.0;
    factor_quarter<0>(d, w, s, Lt, Wt);
    factor_quarter<1>(d, w, s, Lt, Wt);
    factor_quarter<2>(d, w, s, Lt, Wt);
    factor_quarter<3>(d, w, s, Lt, Wt);
    __syncthreads();
    if (threadIdx.x < B) {
        const double p = s.piv[threadIdx.x], rs = rsqrt_f64(p);
        s.rscal[threadIdx.x] = rs;
        s.scal[threadIdx.x] = p * rs;
    }
    __syncthreads();
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
#pragma unroll
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
            if (is_a && p == 1) continue;      // (j + 1, j): the diagonal role of row j + 1 owns it
            if (is_a && i == j) {
                // -- the diagonal block AND the tile to its left, (j, j - 1):
                // the chain FACTOR(j - 1) -> L_j,j-1 -> update of A_jj ->
                // FACTOR(j) stays in one workgroup from the moment
                // L_j-1,j-1^-1 arrives (a role of its own for the tile to the
                // left cost the chain a store, a flag and a staging: 2.7 of
                // 29 us per panel).  Own entries are fetched before the
                // waits: nothing in this launch writes them before this role.
                const int ti = tid >> 4, tj = tid & 15;
                double d[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = ti + 16 * u, c = tj + 16 * v;
                        const int gr = B * j + r, gc = B * j + c;
                        d[u][v] = (gr < n && gc < n) ? ld_sc1(a.A + (size_t)gr * a.ld + gc)
                                                     : (r == c ? 1.0 : 0.0);
                    }
                acc_t T;        // sum_{k < j - 1} L_jk L_j-1,k^T; then A_j,j-1 less it
                T.zero();
                if (j > 0) {
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int gr = B * j + acc_t::row(rb, g), gc = B * (j - 1) + acc_t::col(cb);
                                T.c[rb][cb][g] = gr < n ? -ld_sc1(a.A + (size_t)gr * a.ld + gc) : 0.0;
                            }
                }
                for (int k = 0; k + 1 < j; ++k) {
                    if (!sy.wait(sy.flag_l + j * nb + k, sy.flag_l + (j - 1) * nb + k)) return;
                    stage_tile(P, l_tile(j, k));
                    stage_tile(Q, l_tile(j - 1, k));
                    __syncthreads();
                    S.mma(P, P);
                    T.mma(P, Q);
                    __syncthreads();
                }
                if (j > 0) {
                    // L_j,j-1 = (A_j,j-1 - sum) L_j-1,j-1^-T; T holds the negative
                    T.to_lds(P, -1.0);
                    if (!sy.wait(sy.flag_l + (j - 1) * nb + j - 1, nullptr)) return;
                    stamp(1);
                    stage_tile(Q, tile_src_t{a.Linv + ((size_t)(j - 1) << 12), B, B, B, false});
                    __syncthreads();
                    stamp(2);
                    T.zero();
                    T.mma(P, Q);
                    __syncthreads();
                    double *const out = a.A + (size_t)(B * j) * a.ld + B * (j - 1);
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int r = acc_t::row(rb, g), c = acc_t::col(cb);
                                if (r < rows_of(j)) st_sc1(out + (size_t)r * a.ld + c, T.c[rb][cb][g]);
                            }
                    // ... and its share of A_jj, straight from the registers
                    T.to_lds(P, 1.0);
                    __syncthreads();
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
                // (L_j,j-1 for the other roles of row and column j: its stores
                // have drained behind the product above)
                if (j > 0) sy.publish(sy.flag_l + j * nb + j - 1);
                stamp(4);
                factor_block(d, fs, P, Q);        // (its first barrier is behind these reads of P)
                stamp(5);
                // what the roles below and to the right wait for is L_jj^-1
                double *const li = a.Linv + ((size_t)j << 12);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int e = tid + 256 * q, r = e >> 6, c = e & 63;
                    st_sc1(li + e, c <= r ? Q[r][c] * fs.rscal[r] : 0.0);
                }
                sy.publish(sy.flag_l + j * nb + j);
                stamp(6);
                // (off the critical path: L_jj for the caller and the block's
                // share of log det L)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int e = tid + 256 * q, r = e >> 6, c = e & 63;
                    const int gr = B * j + r, gc = B * j + c;
                    if (gr < n && gc < n)
                        st_sc1(a.A + (size_t)gr * a.ld + gc,
                               c < r ? P[r][c] * fs.scal[c] : (c == r ? fs.scal[c] : 0.0));
                }
                if (tid < 16)
                    red[tid] = log(fs.scal[tid]) + log(fs.scal[tid + 16]) + log(fs.scal[tid + 32]) +
                               log(fs.scal[tid + 48]);
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
