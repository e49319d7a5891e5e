// Round 6: the factorisation of ONE 64 x 64 diagonal block of potrf.hip in
// isolation, timed with the wall clock (100 MHz) per block, with parts of a
// step taken out (compile-time variants): which part of the ~0.5 us per pair
// of columns is the barrier, the reciprocals, the instruction stream?
//   hipcc --offload-arch=gfx950 -O3 -fno-fast-math [-DVARIANT=n] factor_probe.hip
#ifndef VARIANT
#define VARIANT 0
#endif
#if VARIANT == 1      // no barrier between publish and read (wrong values)
#define GD_POTRF_STEP_BARRIER() __builtin_amdgcn_s_waitcnt(0xc07f)
#elif VARIANT == 2    // no reciprocal estimate + refinement
#define GD_POTRF_RCP(x) ((x) * 0.999)
#elif VARIANT == 3    // steps of a quarter unrolled
#define GD_POTRF_STEP_LOOP _Pragma("unroll")
#elif VARIANT == 4    // wave barrier only: no s_barrier, LDS counter drained
#define GD_POTRF_STEP_BARRIER() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#include "../../graphdot_amd/model/gaussian_process/potrf.hip"
#include <cstdio>
#include <vector>

extern "C" __global__ __launch_bounds__(256)
void factor_probe(const double *A, double *out, unsigned long long *t, int reps) {
    __shared__ double P[B][LT];
    __shared__ double Q[B][LT];
    __shared__ factor_lds_t fs;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    double acc = 0;
    const unsigned long long w0 = wall_clock64();
    for (int rep = 0; rep < reps; ++rep) {
        double d[4][4];
        for (int u = 0; u < 4; ++u)
            for (int v = 0; v < 4; ++v) d[u][v] = A[(ti + 16 * u) * B + tj + 16 * v] + rep * 1e-9;
        __syncthreads();
        factor_block(d, fs, P, Q);
        acc += P[(tid >> 2) & 63][tid & 63] + Q[(tid >> 2) & 63][tid & 63] + fs.scal[tid & 63];
    }
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0) t[blockIdx.x] = w1 - w0;
    // the factor of the last repetition, for a check against the host
    if (blockIdx.x == 0)
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            out[65536 + e] = c < r ? P[r][c] * fs.scal[c] : (c == r ? fs.scal[c] : 0.0);
        }
}

int main() {
    std::vector<double> A(B * B), L(B * B);
    for (int r = 0; r < B; ++r)
        for (int c = 0; c < B; ++c) A[r * B + c] = (r == c ? 2.0 : 0.0) + 1.0 / (1.0 + r + c) + 0.01 * ((r * 7 + c * 7) % 5);
    for (int r = 0; r < B; ++r)
        for (int c = r; c < B; ++c) A[c * B + r] = A[r * B + c];
    double *dA, *dout; unsigned long long *dt;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dout, (65536 + 4096) * 8); hipMalloc(&dt, 256 * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const int reps = 200;
    for (int grid : {1, 256}) {
        for (int k = 0; k < 2; ++k) factor_probe<<<grid, 256>>>(dA, dout, dt, reps);
        std::vector<unsigned long long> t(256);
        hipMemcpy(t.data(), dt, 256 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(L.data(), dout + 65536, 4096 * 8, hipMemcpyDeviceToHost);
        // host check: || L L^T - A ||
        double err = 0;
        for (int r = 0; r < B; ++r)
            for (int c = 0; c <= r; ++c) {
                double s = 0;
                for (int k = 0; k <= c; ++k) s += L[r * B + k] * L[c * B + k];
                double e = s - (A[r * B + c] + (reps - 1) * 1e-9);
                if (e < 0) e = -e;
                if (e > err) err = e;
            }
        printf("variant %d, %3d workgroups: %.2f us per block (32 steps: %.0f ns per step), max |L L^T - A| = %.1e\n",
               VARIANT, grid, t[0] * 0.01 / reps, t[0] * 10.0 / reps / 32, err);
    }
    return 0;
}
