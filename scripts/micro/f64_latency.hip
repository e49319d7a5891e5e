// Round 6 micro-benchmark: what a dependent chain of double-precision
// instructions costs one wave on gfx950 (the factorisation of a diagonal
// block in potrf.hip is one such chain per pair of columns), what a
// publish -> barrier -> read round trip through LDS costs four waves, and
// the clock these run at.  hipcc --offload-arch=gfx950 -O3 f64_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 2048

template<int MODE>
__global__ __launch_bounds__(256) void probe(double *out, unsigned long long *t, double x0, double y0) {
    __shared__ double buf[2][256];
    double x = x0 + threadIdx.x * 1e-9, y = y0;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (MODE == 0) {            // dependent v_fma_f64
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, y);
    } else if (MODE == 1) {     // dependent v_mul_f64
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = x * y;
    } else if (MODE == 2) {     // dependent v_rsq_f64
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x);
    } else if (MODE == 3) {     // independent v_fma_f64 (8 chains)
        double a[8];
        for (int k = 0; k < 8; ++k) a[k] = x + k;
#pragma unroll 2
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], y, y);
        x = 0;
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (MODE == 4) {     // LDS write -> barrier -> read of another thread's value
        for (int i = 0; i < N; ++i) {
            buf[i & 1][threadIdx.x] = x;
            __syncthreads();
            x = buf[i & 1][(threadIdx.x + 65) & 255] + y;
        }
    } else if (MODE == 5) {     // dependent v_cndmask pair (a double select)
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = x > 0.0 ? x + y : y;
    } else if (MODE == 6) {     // dependent f32 fma for comparison
        float xf = (float)x, yf = (float)y;
#pragma unroll 16
        for (int i = 0; i < N; ++i) xf = __builtin_fmaf(xf, yf, yf);
        x = xf;
    } else if (MODE == 8) {     // independent v_fma_f64, THREE distinct register sources (acc += b * c)
        double a[8], b[8], c[8];
        for (int k = 0; k < 8; ++k) { a[k] = x + k; b[k] = y + k * 1e-3; c[k] = y - k * 1e-3; }
#pragma unroll 2
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = __builtin_fma(b[k], c[k], a[k]);
                asm volatile("" : "+v"(a[k]), "+v"(b[k]), "+v"(c[k]));
            }
        x = 0;
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (MODE == 9) {     // the same with ONE shared multiplier register (acc += b * y)
        double a[8], b[8];
        for (int k = 0; k < 8; ++k) { a[k] = x + k; b[k] = y + k * 1e-3; }
#pragma unroll 2
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = __builtin_fma(b[k], y, a[k]);
                asm volatile("" : "+v"(a[k]), "+v"(b[k]));
            }
        x = 0;
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (MODE == 10) {    // independent v_mul_f64, two distinct sources
        double a[8], b[8], c[8];
        for (int k = 0; k < 8; ++k) { a[k] = x + k; b[k] = y + k * 1e-3; c[k] = y - k * 1e-3; }
#pragma unroll 2
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = b[k] * c[k];
                asm volatile("" : "+v"(a[k]), "+v"(b[k]), "+v"(c[k]));
            }
        x = 0;
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (MODE == 7) {     // barrier alone
        for (int i = 0; i < N; ++i) { __syncthreads(); x += y; }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

template<int MODE> void run(const char *name, int threads) {
    double *out; unsigned long long *t, h[2];
    hipMalloc(&out, 256 * sizeof(double)); hipMalloc(&t, 16);
    for (int rep = 0; rep < 3; ++rep) probe<MODE><<<1, threads>>>(out, t, 1.0000001, 0.9999999);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-44s threads %3d: %7.1f cycle-counter ticks, %7.2f ns per step (%.0f MHz counter)\n", name, threads,
           (double)h[0] / N, (double)h[1] * 10.0 / N, (double)h[0] / ((double)h[1] * 0.01));
    hipFree(out); hipFree(t);
}

int main() {
    // keep the chip busy for a moment first (clocks)
    for (int threads : {64, 256}) {
        run<0>("dependent v_fma_f64", threads);
        run<1>("dependent v_mul_f64", threads);
        run<2>("dependent v_rsq_f64", threads);
        run<3>("8 independent v_fma_f64 chains (per fma)", threads);
        run<5>("dependent compare + add + select (f64)", threads);
        run<6>("dependent v_fma_f32", threads);
        run<8>("independent fma, 3 distinct sources (per fma)", threads);
        run<9>("independent fma, shared multiplier (per fma)", threads);
        run<10>("independent mul, 2 distinct sources (per mul)", threads);
    }
    run<4>("LDS write -> barrier -> read + add", 256);
    run<7>("barrier + add", 256);
    return 0;
}
