// vrg_lanes.hpp -- the DPP wave shifts of the wave-march kernels (left / right 3x3 taps from the neighbouring lanes)
#pragma once
#include <hip/hip_runtime.h>

namespace vrg {

__device__ __forceinline__ float lane_prev(float v) {   // value held by lane-1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {   // value held by lane+1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// the same shifts for operands of an add (steady rows): no `old` value and bound_ctrl, so that the backend can fold the shift into
// the add's first operand (v_add_f32_dpp) -- the lane at the wave's end reads 0.0, and it is a halo lane whose result is dropped
__device__ __forceinline__ float tap_prev(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float tap_next(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

}  // namespace vrg
