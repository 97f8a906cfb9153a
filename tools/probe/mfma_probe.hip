// Probe: register layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (diagnostic, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[v * 64 + l] = c[v];
}
int main()
{
    float ha[64], hb[64], hd[256], *da, *db, *dd;
    for (int l = 0; l < 64; ++l) { ha[l] = (float)(l + 1); hb[l] = (float)(1000 + l); }
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
            const float expect = ha[4 * (l / 4) + v] * hb[l];  // D_b[i = v][j = l % 4] = A_b[v] * B_b[l % 4]
            if (hd[v * 64 + l] != expect) { ok = 0; if (l < 8) printf("v=%d l=%d got %g expect %g\n", v, l, hd[v * 64 + l], expect); }
        }
    printf("layout D[v][lane] = A[4*(lane/4)+v] * B[lane]: %s\n", ok ? "CONFIRMED" : "MISMATCH");
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l], hd[64 + l], hd[128 + l], hd[192 + l]);
    return 0;
}
