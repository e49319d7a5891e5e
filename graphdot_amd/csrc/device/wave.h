// wave64 primitives for gfx950: lane id, wave-wide sums (DPP within 16-lane
// rows + v_readlane across rows), workgroup sums for multi-wave jobs.
// (Role of the reference's graphdot/cpp/util_cuda.h:8-70, written for
// 64-wide wavefronts.)
#ifndef GRAPHDOT_HIP_WAVE_H_
#define GRAPHDOT_HIP_WAVE_H_
#include <hip/hip_runtime.h>

namespace graphdot {
namespace wave {

constexpr int size = 64;

__device__ __forceinline__ int laneid() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

template<int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(t);
}

template<int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_add_masked(float v) {
    // lanes outside ROW_MASK keep v (old = v, bound_ctrl off)
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true);
    return v + __int_as_float(t);
}

// Sum over the 64 lanes; every lane gets the same (bitwise identical) total.
// 4 DPP steps give every lane its 16-lane row sum; row_bcast15 / row_bcast31
// (gfx9 wave-level DPP) then chain the four rows into lane 63, which is read
// back through an SGPR: 6 VALU + 1 v_readlane.
__device__ __forceinline__ float sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror : 8-lane sums
    v = dpp_add<0x140>(v);  // row_mirror      : 16-lane sums in every lane
    // row_bcast:15 -> rows 1 and 3 add the total of the previous row;
    // row_bcast:31 -> rows 2 and 3 add lane 31 (= rows 0 + 1).  Written as
    // asm: the builtin form costs an extra v_mov (old value) + v_add each.
    // (s_nop 1: two wait states between a VALU write and a DPP read of it.)
    // (not volatile: a sum nobody reads must stay removable)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc\n\t"
                 "s_nop 0"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// 64-bit DPP move: both halves with the same control
template<int CTRL, int ROW_MASK, bool BOUND> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, BOUND);
    const int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, BOUND);
    return __hiloint2double(thi, tlo);
}

// Same reduction tree in double: v_add_f64 takes no DPP operand, so every
// step is two v_mov_dpp + one v_add_f64 (20 instructions in all) -- against a
// __shfl_xor butterfly, which on gfx950 is two ds_bpermute through the LDS
// crossbar per step, i.e. six dependent LDS round trips per sum.
__device__ __forceinline__ double sum(double v) {
    v += dpp_mov<0xB1, 0xF, true>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xF, true>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xF, true>(v);   // row_half_mirror
    v += dpp_mov<0x140, 0xF, true>(v);   // row_mirror: 16-lane sums in every lane
    // row_bcast:15: every row adds the total of the previous one (row 0: no
    // source lane, bound_ctrl reads 0); row_bcast:31: rows 2 and 3 add lane 31
    // (= rows 0 + 1).  All rows enabled: lane 63, the only one read, is right
    // -- r3 + r2 + (r1 + r0) -- and the destination needs no "old" value: with
    // rows masked off the moves cost two v_mov 0 each.
    v += dpp_mov<0x142, 0xF, true>(v);
    v += dpp_mov<0x143, 0xF, true>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Two wave sums for little more than the price of one.  v_permlane32_swap
// exchanges the upper half of its first operand with the lower half of its
// second, so after one swap and one add lanes 0-31 hold a(l) + a(l + 32) and
// lanes 32-63 hold b(l - 32) + b(l); a 32-lane reduction (4 DPP steps + one
// row_bcast:15) then leaves sum(a) in lane 31 and sum(b) in lane 63.
// float: 7 VALU + 2 v_readlane instead of 12 + 2; double: 20 instead of 40.
__device__ __forceinline__ void sum2(float &a, float &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\ts_nop 0" : "+v"(v));
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ void sum2(double &a, double &b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    double v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    v += dpp_mov<0xB1, 0xF, true>(v);
    v += dpp_mov<0x4E, 0xF, true>(v);
    v += dpp_mov<0x141, 0xF, true>(v);
    v += dpp_mov<0x140, 0xF, true>(v);
    v += dpp_mov<0x142, 0xF, true>(v);     // (lanes 31 and 63: rows 1 and 3)
    const int l = __double2loint(v), h = __double2hiint(v);
    a = __hiloint2double(__builtin_amdgcn_readlane(h, 31), __builtin_amdgcn_readlane(l, 31));
    b = __hiloint2double(__builtin_amdgcn_readlane(h, 63), __builtin_amdgcn_readlane(l, 63));
}

}  // namespace wave
}  // namespace graphdot
#endif
