// Probe: quad_transpose via DPP quad_perm on gfx950 (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline f32x4 quad_transpose(f32x4 m, int lane)
{
    const bool hi2 = lane & 2, hi1 = lane & 1;
    auto qp = [](float v, auto ctrl) -> float {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    using X2 = std::integral_constant<int, 0x4e>;
    using X1 = std::integral_constant<int, 0xb1>;
    // (every DPP move is executed by ALL lanes, then selected: inside a conditional it would read inactive lanes)
    const float p0 = qp(m[0], X2{}), p1 = qp(m[1], X2{}), p2 = qp(m[2], X2{}), p3 = qp(m[3], X2{});
    f32x4 a;
    a[0] = hi2 ? p2 : m[0];
    a[2] = hi2 ? m[2] : p0;
    a[1] = hi2 ? p3 : m[1];
    a[3] = hi2 ? m[3] : p1;
    const float q0 = qp(a[0], X1{}), q1 = qp(a[1], X1{}), q2 = qp(a[2], X1{}), q3 = qp(a[3], X1{});
    f32x4 o;
    o[0] = hi1 ? q1 : a[0];
    o[1] = hi1 ? a[1] : q0;
    o[2] = hi1 ? q3 : a[2];
    o[3] = hi1 ? a[3] : q2;
    return o;
}
__global__ void probe(float *d)
{
    const int l = threadIdx.x;
    f32x4 m = {l * 10.f + 0, l * 10.f + 1, l * 10.f + 2, l * 10.f + 3};
    f32x4 o = quad_transpose(m, l);
    for (int v = 0; v < 4; ++v) d[v * 64 + l] = o[v];
}
int main()
{
    float hd[256], *dd;
    (void)hipMalloc(&dd, 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dd);
    (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
            const int src_lane = (l & ~3) + v, comp = l & 3;  // o[v](lane p) = m[p](lane v)
            const float expect = src_lane * 10.f + comp;
            if (hd[v * 64 + l] != expect) { ok = 0; if (l < 8) printf("v=%d l=%d got %g expect %g\n", v, l, hd[v * 64 + l], expect); }
        }
    printf("quad_transpose: %s\n", ok ? "CONFIRMED" : "MISMATCH");
    return 0;
}
