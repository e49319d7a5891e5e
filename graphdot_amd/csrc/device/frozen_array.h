// Read-only view of a variable-length attribute (ptr + length) embedded in a
// node/edge struct; layout {T const*; int32} matches the packer's
// FrozenArray dtype (kernel/marginalized/_devicegraph.py), same as the
// reference's graphdot/cpp/frozen_array.h:12-44.
#ifndef GRAPHDOT_HIP_FROZEN_ARRAY_H_
#define GRAPHDOT_HIP_FROZEN_ARRAY_H_
#include <hip/hip_runtime.h>
#include "numpy_type.h"

namespace graphdot {
namespace numpy_type {

template<class T> struct frozen_array {
    using element_type = T;
    T const *_data = nullptr;
    int32 size = 0;

    __host__ __device__ T const *begin() const { return _data; }
    __host__ __device__ T const *end() const { return _data + size; }
    __host__ __device__ T const &operator[](int i) const { return _data[i]; }
};

}  // namespace numpy_type
}  // namespace graphdot
#endif
