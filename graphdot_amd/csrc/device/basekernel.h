// Helpers called by generated microkernel expressions: normalize(),
// normalize_jacobian(), convolution<mean>(), convolution_jacobian<mean>(),
// dotproduct().  Semantics follow the reference's
// graphdot/cpp/basekernel/{normalize,convolution,dotproduct}.h; arithmetic is
// done in graphdot::real_t.
#ifndef GRAPHDOT_HIP_BASEKERNEL_H_
#define GRAPHDOT_HIP_BASEKERNEL_H_
#include <hip/hip_runtime.h>
#include "fmath.h"
#include "frozen_array.h"

namespace graphdot {
namespace basekernel {

template<class F, class X, class Y>
__device__ __forceinline__ real_t normalize(F const f, X const &x, Y const &y) {
    real_t const kxx = f(x, x);
    real_t const kyy = f(y, y);
    real_t const s = kxx * kyy;
    return s > 0 ? real_t(f(x, y)) * graphdot::rsqrt(s) : real_t(0);
}

template<class F, class J, class X, class Y>
__device__ __forceinline__ real_t normalize_jacobian(F const f, J const j, X const &x, Y const &y) {
    real_t const kxx = f(x, x), kxy = f(x, y), kyy = f(y, y);
    real_t const jxx = j(x, x), jxy = j(x, y), jyy = j(y, y);
    real_t const s = kxx * kyy;
    if (s > 0) {
        real_t const rs = graphdot::rsqrt(s);
        return jxy * rs - real_t(0.5) * kxy * rs * rs * rs * (jxx * kyy + kxx * jyy);
    }
    return real_t(0);
}

template<bool mean, class F, class X, class Y>
__device__ __forceinline__ real_t convolution(F const f, X const &x, Y const &y) {
    real_t k = 0;
    for (auto const &_1 : x)
        for (auto const &_2 : y) k += f(_1, _2);
    return mean ? k / real_t(x.size * y.size) : k;
}

template<bool mean, class J, class X, class Y>
__device__ __forceinline__ real_t convolution_jacobian(J const j, X const &x, Y const &y) {
    real_t dk = 0;
    for (auto const &_1 : x)
        for (auto const &_2 : y) dk += j(_1, _2);
    return mean ? dk / real_t(x.size * y.size) : dk;
}

template<class T>
__device__ __forceinline__ real_t dotproduct(numpy_type::frozen_array<T> const &x,
                                             numpy_type::frozen_array<T> const &y) {
    real_t sum = 0;
    for (int i = 0; i < x.size; ++i) sum += real_t(x._data[i]) * real_t(y._data[i]);
    return sum;
}

}  // namespace basekernel
}  // namespace graphdot
#endif
