// Lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, found by experiment: A = 1 in one lane, B = 1 in one lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(int la, int lb, float *out)
{
    const int l = threadIdx.x;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(l == la ? 1.0f : 0.0f, l == lb ? 1.0f : 0.0f, d, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = d[v];
}
// issue cost: a chain-free loop of 4x4x1 (2 accumulators alternating / 8 accumulators) and of 16x16x4, cycles per instruction
template <int MODE>
__global__ void rate(float *out, unsigned long long *cyc)
{
    f32x4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)threadIdx.x, b = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) d[i & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[i & 1], 0, 0, 0);
            if (MODE == 1) d[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[i], 0, 0, 0);
            if (MODE == 2) d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[i], 0, 0, 0);
            if (MODE == 3) d[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[0], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += d[i][0] + d[i][1] + d[i][2] + d[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// cross-row reduce-scatter with the gfx950 lane swaps: four registers, each holding a value per lane, become one
// register whose row r (16 lanes) holds, lane by lane, the sum over the four rows of register {0, 2, 1, 3}[r]
__global__ void swaps(const float *in, float *out)
{
    const int l = threadIdx.x;
    float a = in[l], b = in[64 + l], c = in[128 + l], d = in[192 + l];
    auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float ab = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
    auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
    const float cd = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
    auto r3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ab), __float_as_uint(cd), false, false);
    out[l] = __uint_as_float(r3[0]) + __uint_as_float(r3[1]);
}
int main()
{
    {
        float h[256], o[64], *di, *dо;
        for (int i = 0; i < 256; ++i) h[i] = (float)((i / 64 + 1) * 1000 + (i % 64));
        hipMalloc(&di, sizeof h); hipMalloc(&dо, sizeof o);
        hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
        swaps<<<1, 64>>>(di, dо);
        hipMemcpy(o, dо, sizeof o, hipMemcpyDeviceToHost);
        int bad = 0;
        const int vmap[4] = {0, 2, 1, 3};
        for (int l = 0; l < 64; ++l) {
            const int v = vmap[l / 16];
            float want = 0;
            for (int r = 0; r < 4; ++r) want += h[v * 64 + r * 16 + (l % 16)];
            if (o[l] != want) ++bad;
        }
        printf("lane-swap reduce-scatter: %d mismatches (row r holds register {0,2,1,3}[r])\n", bad);
    }
    {
        float *o; unsigned long long *c, h;
        hipMalloc(&o, 1024); hipMalloc(&c, 8);
        const char *nm[4] = {"4x4x1, 2 chains", "4x4x1, 8 chains", "16x16x4, 8 chains", "4x4x1, 1 chain"};
        for (int m = 0; m < 4; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) rate<0><<<1, 64>>>(o, c);
                if (m == 1) rate<1><<<1, 64>>>(o, c);
                if (m == 2) rate<2><<<1, 64>>>(o, c);
                if (m == 3) rate<3><<<1, 64>>>(o, c);
                hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            }
            printf("%s: %.2f memtime ticks per instruction (memtime runs at 100 MHz: x clock/100MHz for cycles)\n", nm[m], (double)h / 8000.0);
        }
    }
    float *d;
    hipMalloc(&d, 256 * 4);
    float h[256];
    for (int la : {0, 1, 5, 62})
        for (int lb : {0, 2, 4, 6, 63}) {
            k<<<1, 64>>>(la, lb, d);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("A lane %d, B lane %d ->", la, lb);
            for (int i = 0; i < 256; ++i)
                if (h[i] != 0.f) printf(" (lane %d, vgpr %d)=%g", i / 4, i % 4, h[i]);
            printf("\n");
        }
    return 0;
}
