// Scalar math helpers referenced by generated microkernel expressions:
// graphdot::ipow<N>, graphdot::ripow<N> (integer powers by repeated squaring,
// same results as the reference's graphdot/cpp/fmath.h:8-33) and
// graphdot::rsqrt, graphdot::pow / log / exp.
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
// (double: under -ffast-math `1.0 / sqrt(x)` becomes a bare v_rsq_f64, good
// to ~1e-8 only; two Newton steps on the hardware estimate restore double
// precision)
// x = 0 and x = +inf keep the hardware result (inf, 0) like 1 / sqrt(x): the
// refinement would turn them into 0 * inf * inf = NaN.  (v_cmp_class: an
// `x == 0 || isinf(x)` test does not survive -ffast-math.)
__device__ __forceinline__ double rsqrt(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double y = y0 * (1.5 - h * y0 * y0);
    y = y * (1.5 - h * y * y);
    return __builtin_amdgcn_class(x, 0x20 | 0x40 | 0x200) ? y0 : y;   // -0, +0, +inf
}

// pow / log / exp in the arithmetic of their arguments.  The code generator
// prints the reference's fast-math float spellings (__powf, __logf, __expf:
// microkernel/_base.py, `k**c`); the backend rewrites them to these overloads
// (codegen/sympy_printer.py::to_real_expr), so that a float build takes the
// gfx950 hardware transcendentals (v_log_f32 / v_exp_f32 under -ffast-math)
// and a double build stays in double.
__device__ __forceinline__ float pow(float x, float y) { return __builtin_powf(x, y); }
// (double: the OCML routines of the HIP math header -- the gfx950 backend has
// no selection for the fast-math f64 llvm.pow / llvm.log intrinsics)
__device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
template<class A, class B> __device__ __forceinline__ auto pow(A x, B y) {
    using T = decltype(x + y);
    return graphdot::pow(T(x), T(y));
}
__device__ __forceinline__ float log(float x) { return __builtin_logf(x); }
__device__ __forceinline__ double log(double x) { return ::log(x); }
// (exp(x) as exp2(x * log2 e) spelt out: the backend expands llvm.exp to
// exactly this, but only at instruction selection -- written here, fast-math
// reassociation folds log2 e into a loop-invariant factor of the argument,
// e.g. -0.5 / length_scale^2 of SquareExponential: one multiplication per
// evaluation less)
__device__ __forceinline__ float exp(float x) { return __builtin_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ double exp(double x) { return ::exp(x); }

// Two evaluations of a microkernel at once: the generated functors are
// templates of their argument type, and a record whose 4-byte leaves are
// replaced by pairs (`pk2<float>`, `pk2<int32>`: the backend prints such an
// `edge2_t` next to `edge_t` when every leaf qualifies and the expression only
// calls what is overloaded here) evaluates the same expression on two records
// with the packed float instructions of gfx950 (v_pk_add_f32, v_pk_mul_f32,
// v_pk_fma_f32: two lanes' worth per issue).  Comparisons, ?: with scalar
// branches and arithmetic with scalar hyperparameters work on these vector
// types as they are.
template<class T> using pk2 = T __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2<float> exp(pk2<float> x) {
    const pk2<float> y = x * 1.44269504088896340736f;
    return pk2<float>{__builtin_exp2f(y.x), __builtin_exp2f(y.y)};
}
// (element k of a packed or a plain result)
template<class T> __device__ __forceinline__ T lane_of(pk2<T> v, int k) { return k ? v.y : v.x; }
__device__ __forceinline__ float lane_of(float v, int) { return v; }
__device__ __forceinline__ float lane_of(int v, int) { return (float)v; }

}  // namespace graphdot

#endif
