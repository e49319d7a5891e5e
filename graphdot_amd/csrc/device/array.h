// Fixed-size register array with a well-defined zero-length specialisation
// (generated Jacobians may have no entries).  API of graphdot/cpp/array.h.
#ifndef GRAPHDOT_HIP_ARRAY_H_
#define GRAPHDOT_HIP_ARRAY_H_
#include <hip/hip_runtime.h>

namespace graphdot {

template<class T, int N> struct array {
    using element_type = T;
    constexpr static int size = N;
    T _data[N];

    __host__ __device__ array() = default;
    template<class U> __host__ __device__ __forceinline__ array(U const value) {
#pragma unroll
        for (int i = 0; i < N; ++i) _data[i] = value;
    }
    __host__ __device__ __forceinline__ T &operator[](int i) { return _data[i]; }
    __host__ __device__ __forceinline__ T const &operator[](int i) const { return _data[i]; }
};

template<class T> struct array<T, 0> {
    using element_type = T;
    constexpr static int size = 0;
    __host__ __device__ array() = default;
    template<class U> __host__ __device__ array(U const) {}
    __host__ __device__ __forceinline__ T operator[](int) const { return T{}; }
};

}  // namespace graphdot
#endif
