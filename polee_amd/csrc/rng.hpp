// Counter-based normal noise shared by the VI loop (vi.hip) and the regression model (regression.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace polee {

// ---- counter-based RNG: Philox4x32-10, one N(0,1) per (seed, step, draw, k) -------------
__host__ __device__ inline void philox_round(uint32_t (&c)[4], const uint32_t (&k)[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}
// Four N(0,1) per Philox block: counter (k, draw / 4, step), two Box-Muller pairs.
__device__ inline void philox_randn4(uint64_t seed, uint32_t step, uint32_t group, uint32_t k, float (&z)[4])
{
    uint32_t c[4] = {k, group, step, 0x706f6c65u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, key);
        key[0] += 0x9E3779B9u;
        key[1] += 0xBB67AE85u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // Box-Muller on two 24-bit uniforms in (0,1), on the hardware's transcendental units: v_log_f32 (log2), v_sqrt_f32 and
        // v_sin_f32 / v_cos_f32, which take their argument in revolutions -- sin(2 pi u2) is v_sin_f32(u2), no range reduction.
        // (~1e-6 absolute on a draw: noise, not arithmetic the reference defines; libm's logf + sincosf were ~250 instructions
        // per node of the VI update, a tenth of the kernel)
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __log2f(u1));  // sqrt(-2 ln u1)
        z[2 * h] = r * __builtin_amdgcn_cosf(u2);
        z[2 * h + 1] = r * __builtin_amdgcn_sinf(u2);
    }
}
__device__ inline float philox_randn(uint64_t seed, uint32_t step, uint32_t draw, uint32_t k)
{
    float z[4];
    philox_randn4(seed, step, draw >> 2, k, z);
    return z[draw & 3];
}

}  // namespace polee
