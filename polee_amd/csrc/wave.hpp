// Wave64 reductions with DPP (no LDS traffic, unlike __shfl): shared by loglik.hip and regression.hip.
// DPP reads of inactive lanes return 0 / stale data: call these with all 64 lanes active.
#pragma once
#include <hip/hip_runtime.h>

namespace polee {

// ---- wave-level sum via DPP (gfx9 row_shr / row_bcast), result valid in lane 63 ------------
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ inline float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __int_as_float(moved);
}
__device__ inline float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111, 0xf, 0xf>(v);  // row_shr:1   (inclusive scan inside each row of 16 lanes)
    v = dpp_add<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row total
    v = dpp_add<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}
// the same for N values at once, step-major so that the N dependency chains interleave
template <int N>
__device__ inline void wave_sum_to_lane63_n(float (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x111, 0xf, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x112, 0xf, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x114, 0xf, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x118, 0xf, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x142, 0xa, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x143, 0xc, 0xf>(v[i]);
}

}  // namespace polee
