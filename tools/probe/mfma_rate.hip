// Probe: issue cost of v_mfma_f32_4x4x1_16b_f32 vs v_mfma_f32_16x16x4_f32 on gfx950 (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float *out, int iters)
{
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 4x4x1, 4 independent chains
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
        } else if (MODE == 1) {  // 4x4x1, one dependent chain
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
        } else if (MODE == 2) {  // 16x16x4, 4 independent chains
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        } else {  // 4x4x1, two chains
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = a0 + a1 + a2 + a3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (4.0f * iters);
}
template <int MODE>
void run(const char *name, int waves_per_simd)
{
    float *d, h;
    (void)hipMalloc(&d, 1 << 20);
    const int iters = 4096;
    // one CU worth: waves_per_simd * 4 waves in one block of that many * 64 threads (max 1024 threads)
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, d, iters);
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-28s %d wave(s)/SIMD: %.1f s_memtime ticks per MFMA per wave\n", name, waves_per_simd, h);
    (void)hipFree(d);
}
int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("4x4x1 x4 independent", w);
        run<1>("4x4x1 dependent chain", w);
        run<3>("4x4x1 two chains", w);
        run<2>("16x16x4 x4 independent", w);
    }
    return 0;
}
