// Scalar math helpers referenced by generated microkernel expressions:
// graphdot::ipow<N>, graphdot::ripow<N> (integer powers by repeated squaring,
// same results as the reference's graphdot/cpp/fmath.h:8-33) and
// graphdot::rsqrt.  The float spellings the code generator emits for a
// --use_fast_math CUDA build (__powf, __logf, rsqrtf) map onto the gfx950
// hardware transcendentals.
#ifndef GRAPHDOT_HIP_FMATH_H_
#define GRAPHDOT_HIP_FMATH_H_
#include <hip/hip_runtime.h>

namespace graphdot {

template<int E, class F> __host__ __device__ constexpr inline F ipow(F base) {
    if constexpr (E == 0) {
        return F(1);
    } else if constexpr (E == 1) {
        return base;
    } else {
        F h = ipow<E / 2>(base);
        return (E % 2) ? h * h * base : h * h;
    }
}

template<int E, class F> __host__ __device__ constexpr inline F ripow(F base) {
    return ipow<E>(F(1) / base);
}

__device__ __forceinline__ float rsqrt(float x) { return __frsqrt_rn(x); }
__device__ __forceinline__ double rsqrt(double x) { return 1.0 / sqrt(x); }

}  // namespace graphdot

#ifndef __powf
#define __powf(x, y) __builtin_powf((x), (y))
#endif
#ifndef __logf
#define __logf(x) __builtin_logf((x))
#endif
#endif
