// Element-wise reparameterisations as standalone entry points (host-pointer API): the Julia functions of
// src/logitnormal.jl:8-55, src/sinh_arcsinh.jl:10-38 and src/kumaraswamy.jl:27-78.  Inside the VI loop the first
// two are fused into vi_sample / vi_update (vi_fused.hpp); these exist so that callers of the individual
// reference functions (alt approximations, tests) find them behind the same C ABI.
#include "common.hpp"
#include "scan.hpp"

namespace polee {

// ---- logit-normal --------------------------------------------------------------------------------------
__global__ void logit_normal_kernel(const float *mu, const float *sigma, const float *zs, int64_t len, double *ys,
                                    double *ladj)
{
    __shared__ double smd[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double l = 0.0;
    if (i < len) {
        const double y = (double)(1.0f / (1.0f + expf(-(mu[i] + zs[i] * sigma[i]))));  // logitnormal.jl:2,12
        ys[i] = y;
        if (ladj) l = log((double)sigma[i] * y * (1 - y));  // :15
    }
    if (ladj) {
        l = block_sum_f64(l, smd);
        if (threadIdx.x == 0) atomicAdd(ladj, l);
    }
}

// logit_normal_transform_gradients! (8-argument form, logitnormal.jl:38-55); z_grad may be null (7-argument form)
__global__ void logit_normal_grad_kernel(const float *zs, const double *ys, const float *sigma, const float *y_grad,
                                         int64_t len, float *z_grad, float *mu_grad, float *sigma_grad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const double y = ys[i], d = y * (1 - y);
    float mg = mu_grad[i], sg = sigma_grad[i];
    mg = (float)((double)mg + d * (double)y_grad[i]);
    sg = (float)((double)sg + d * (double)zs[i] * (double)y_grad[i]);
    mg = (float)((double)mg + (1 - 2 * y));
    sg = (float)((double)sg + ((double)(1.0f / sigma[i]) + (double)zs[i] * (1 - 2 * y)));
    mu_grad[i] = mg;
    sigma_grad[i] = sg;
    if (z_grad) {
        float zg = z_grad[i];
        zg = (float)((double)zg + d * (double)sigma[i] * (double)y_grad[i]);
        zg = (float)((double)zg + (double)sigma[i] * (1 - 2 * y));
        z_grad[i] = zg;
    }
}

// ---- sinh-arcsinh ----------------------------------------------------------------------------------------
__global__ void sinh_asinh_kernel(const float *alpha, const float *zs0, int64_t len, float *zs, double *ladj)
{
    __shared__ double smd[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double l = 0.0;
    if (i < len) {
        const float c = alpha[i] + asinhf(zs0[i]);
        zs[i] = sinhf(c);
        if (ladj) l = (double)logf(coshf(c)) - 0.5 * (double)log1pf(zs0[i] * zs0[i]);  // sinh_arcsinh.jl:18
    }
    if (ladj) {  // a proper reduction (the reference's threaded += is a data race)
        l = block_sum_f64(l, smd);
        if (threadIdx.x == 0) atomicAdd(ladj, l);
    }
}
__global__ void sinh_asinh_grad_kernel(const float *zs0, const float *alpha, const float *z_grad, int64_t len,
                                       float *alpha_grad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const float c = alpha[i] + asinhf(zs0[i]);
    float ag = alpha_grad[i];
    ag += coshf(c) * z_grad[i];  // sinh_arcsinh.jl:32-33
    ag += tanhf(c);              // :36
    alpha_grad[i] = ag;
}

// ---- Kumaraswamy -----------------------------------------------------------------------------------------
__global__ void kumaraswamy_kernel(const float *as, const float *bs, const float *zs, int64_t len, double *ys,
                                   double *ladj)
{
    __shared__ double smd[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double l = 0.0;
    if (i < len) {
        const double a = as[i], b = bs[i], z = zs[i];
        const double ia = 1 / a, ib = 1 / b;
        const double c = 1 - pow(1 - z, ib);  // kumaraswamy.jl:39
        ys[i] = pow(c, ia);
        if (ladj) l = (ib - 1) * log(1 - z) + (ia - 1) * log(c) - log(a * b);  // :44
    }
    if (ladj) {
        l = block_sum_f64(l, smd);
        if (threadIdx.x == 0) atomicAdd(ladj, l);
    }
}
__global__ void kumaraswamy_grad_kernel(const float *zs, const float *as, const float *bs, const float *y_grad,
                                        int64_t len, float *a_grad, float *b_grad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const double a = as[i], b = bs[i], z = zs[i];
    const double ia = 1 / a, ib = 1 / b;
    const double omz_ib = pow(1 - z, ib);
    const double c = 1 - omz_ib, log_c = log(c), log_omz = log(1 - z);
    float ag = a_grad[i], bg = b_grad[i];
    ag = (float)((double)ag + (-log_c / (a * a) - ia));                                                    // :66
    bg = (float)((double)bg + (-log_omz / (b * b) + (ia - 1) * (1 / c) * omz_ib * log_omz / (b * b) - ib));  // :67-69
    ag = (float)((double)ag + (-pow(c, ia) * log_c / (a * a)) * (double)y_grad[i]);                          // :72-73
    bg = (float)((double)bg + (pow(c, ia - 1) * omz_ib * log_omz / (a * b * b)) * (double)y_grad[i]);        // :75-76
    a_grad[i] = ag;
    b_grad[i] = bg;
}

// ---- gene-level non-informative prior (likelihood.jl:114-159), K draws at once -----------------------------
// pass 1: per-gene sums c[g][k] = sum_{i in g} xls[k][i] and member counts; sum_i xs/efflen per draw
__global__ void gene_prior_sums_kernel(const float *efflens, const float *xls, const float *xs, const int32_t *gene_of,
                                       int64_t n, int K, double *gene_c, int *gene_k, double *xsum)
{
    __shared__ double smd[4];
    const int k = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double q = 0.0;
    if (i < n) {
        const int g = gene_of[i];
        if (g >= 0) {
            atomicAdd(&gene_c[(size_t)g * K + k], (double)xls[(size_t)k * n + i]);
            if (k == 0) atomicAdd(&gene_k[g], 1);
        }
        q = (double)(xs[(size_t)k * n + i] / efflens[i]);  // f32 quotient (:140)
    }
    q = block_sum_f64(q, smd);
    if (threadIdx.x == 0) atomicAdd(&xsum[k], q);
}
// pass 2: offdiag[k] = sum_i -xl_grad_i * xls_i with xl_grad_i = -(k_g - 1) / c_g (:130-147)
__global__ void gene_prior_offdiag_kernel(const float *xls, const int32_t *gene_of, int64_t n, int K,
                                          const double *gene_c, const int *gene_k, double *offdiag)
{
    __shared__ double smd[4];
    const int k = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double q = 0.0;
    if (i < n) {
        const int g = gene_of[i];
        if (g >= 0 && gene_k[g] > 1) q = (double)(gene_k[g] - 1) / gene_c[(size_t)g * K + k] * (double)xls[(size_t)k * n + i];
    }
    q = block_sum_f64(q, smd);
    if (threadIdx.x == 0) atomicAdd(&offdiag[k], q);
}
// pass 3: x_grad_i += xl_grad_i * (1/efflen_i) / xsum + (1/efflen_i) * offdiag / xsum^2 (:149-155)
__global__ void gene_prior_apply_kernel(const float *efflens, const int32_t *gene_of, int64_t n, int K,
                                        const double *gene_c, const int *gene_k, const double *xsum,
                                        const double *offdiag, double *x_grad)
{
    const int k = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = gene_of[i];
    const double xlg = (g >= 0 && gene_k[g] > 1) ? -(double)(gene_k[g] - 1) / gene_c[(size_t)g * K + k] : 0.0;
    const double inv_l = (double)(1.0f / efflens[i]);
    const double s = xsum[k];
    x_grad[(size_t)k * n + i] += xlg * (inv_l / s) + inv_l * (offdiag[k] / (s * s));
}

}  // namespace polee

using namespace polee;

namespace {
struct Scratch {
    DevBuf<float> f[6];
    DevBuf<double> d[2];
};
inline unsigned blocks(int64_t len) { return (unsigned)ceil_div(len, 256); }
}  // namespace

namespace polee {
__global__ void fast_exp_kernel(double *x, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) x[i] = fast_exp(x[i]);
}
__global__ void fast_log_kernel(double *x, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) x[i] = fast_log(x[i]);
}
}  // namespace polee

extern "C" {

polee_status polee_logit_normal_transform(polee_ctx *ctx, const float *mu, const float *sigma, const float *zs,
                                          int64_t len, double *ys, double *ladj)
{
    POLEE_TRY(use_device(ctx));
    if (!mu || !sigma || !zs || !ys || len < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (len == 0) return POLEE_OK;
    Scratch s;
    POLEE_TRY(s.f[0].upload(ctx, mu, len));
    POLEE_TRY(s.f[1].upload(ctx, sigma, len));
    POLEE_TRY(s.f[2].upload(ctx, zs, len));
    POLEE_TRY(s.d[0].alloc(ctx, len));
    POLEE_TRY(s.d[1].alloc(ctx, 1));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(s.d[1].p, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(logit_normal_kernel, dim3(blocks(len)), dim3(256), 0, ctx->stream, s.f[0].p, s.f[1].p, s.f[2].p,
                       len, s.d[0].p, ladj ? s.d[1].p : nullptr);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(s.d[0].download(ctx, ys, len));
    if (ladj) POLEE_TRY(s.d[1].download(ctx, ladj, 1));
    return POLEE_OK;
}

polee_status polee_logit_normal_transform_gradients(polee_ctx *ctx, const float *zs, const double *ys,
                                                    const float *sigma, const float *y_grad, int64_t len,
                                                    float *z_grad_or_null, float *mu_grad, float *sigma_grad)
{
    POLEE_TRY(use_device(ctx));
    if (!zs || !ys || !sigma || !y_grad || !mu_grad || !sigma_grad || len < 0)
        return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (len == 0) return POLEE_OK;
    Scratch s;
    POLEE_TRY(s.f[0].upload(ctx, zs, len));
    POLEE_TRY(s.d[0].upload(ctx, ys, len));
    POLEE_TRY(s.f[1].upload(ctx, sigma, len));
    POLEE_TRY(s.f[2].upload(ctx, y_grad, len));
    POLEE_TRY(s.f[3].upload(ctx, mu_grad, len));
    POLEE_TRY(s.f[4].upload(ctx, sigma_grad, len));
    if (z_grad_or_null) POLEE_TRY(s.f[5].upload(ctx, z_grad_or_null, len));
    hipLaunchKernelGGL(logit_normal_grad_kernel, dim3(blocks(len)), dim3(256), 0, ctx->stream, s.f[0].p, s.d[0].p,
                       s.f[1].p, s.f[2].p, len, z_grad_or_null ? s.f[5].p : nullptr, s.f[3].p, s.f[4].p);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(s.f[3].download(ctx, mu_grad, len));
    POLEE_TRY(s.f[4].download(ctx, sigma_grad, len));
    if (z_grad_or_null) POLEE_TRY(s.f[5].download(ctx, z_grad_or_null, len));
    return POLEE_OK;
}

polee_status polee_sinh_asinh_transform(polee_ctx *ctx, const float *alpha, const float *zs0, int64_t len, float *zs,
                                        double *ladj)
{
    POLEE_TRY(use_device(ctx));
    if (!alpha || !zs0 || !zs || len < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (len == 0) return POLEE_OK;
    Scratch s;
    POLEE_TRY(s.f[0].upload(ctx, alpha, len));
    POLEE_TRY(s.f[1].upload(ctx, zs0, len));
    POLEE_TRY(s.f[2].alloc(ctx, len));
    POLEE_TRY(s.d[1].alloc(ctx, 1));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(s.d[1].p, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(sinh_asinh_kernel, dim3(blocks(len)), dim3(256), 0, ctx->stream, s.f[0].p, s.f[1].p, len,
                       s.f[2].p, ladj ? s.d[1].p : nullptr);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(s.f[2].download(ctx, zs, len));
    if (ladj) POLEE_TRY(s.d[1].download(ctx, ladj, 1));
    return POLEE_OK;
}

polee_status polee_sinh_asinh_transform_gradients(polee_ctx *ctx, const float *zs0, const float *alpha,
                                                  const float *z_grad, int64_t len, float *alpha_grad)
{
    POLEE_TRY(use_device(ctx));
    if (!zs0 || !alpha || !z_grad || !alpha_grad || len < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (len == 0) return POLEE_OK;
    Scratch s;
    POLEE_TRY(s.f[0].upload(ctx, zs0, len));
    POLEE_TRY(s.f[1].upload(ctx, alpha, len));
    POLEE_TRY(s.f[2].upload(ctx, z_grad, len));
    POLEE_TRY(s.f[3].upload(ctx, alpha_grad, len));
    hipLaunchKernelGGL(sinh_asinh_grad_kernel, dim3(blocks(len)), dim3(256), 0, ctx->stream, s.f[0].p, s.f[1].p,
                       s.f[2].p, len, s.f[3].p);
    POLEE_KERNEL_CHECK(ctx);
    return s.f[3].download(ctx, alpha_grad, len);
}

polee_status polee_kumaraswamy_transform(polee_ctx *ctx, const float *as, const float *bs, const float *zs, int64_t len,
                                         double *ys, double *ladj)
{
    POLEE_TRY(use_device(ctx));
    if (!as || !bs || !zs || !ys || len < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (len == 0) return POLEE_OK;
    Scratch s;
    POLEE_TRY(s.f[0].upload(ctx, as, len));
    POLEE_TRY(s.f[1].upload(ctx, bs, len));
    POLEE_TRY(s.f[2].upload(ctx, zs, len));
    POLEE_TRY(s.d[0].alloc(ctx, len));
    POLEE_TRY(s.d[1].alloc(ctx, 1));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(s.d[1].p, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(kumaraswamy_kernel, dim3(blocks(len)), dim3(256), 0, ctx->stream, s.f[0].p, s.f[1].p, s.f[2].p,
                       len, s.d[0].p, ladj ? s.d[1].p : nullptr);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(s.d[0].download(ctx, ys, len));
    if (ladj) {
        POLEE_TRY(s.d[1].download(ctx, ladj, 1));
        if (!std::isfinite(*ladj)) return fail(ctx, POLEE_ERR_NONFINITE, "kumaraswamy ladj is not finite (kumaraswamy.jl:49)");
    }
    return POLEE_OK;
}

polee_status polee_kumaraswamy_transform_gradients(polee_ctx *ctx, const float *zs, const float *as, const float *bs,
                                                   const float *y_grad, int64_t len, float *a_grad, float *b_grad)
{
    POLEE_TRY(use_device(ctx));
    if (!zs || !as || !bs || !y_grad || !a_grad || !b_grad || len < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (len == 0) return POLEE_OK;
    Scratch s;
    POLEE_TRY(s.f[0].upload(ctx, zs, len));
    POLEE_TRY(s.f[1].upload(ctx, as, len));
    POLEE_TRY(s.f[2].upload(ctx, bs, len));
    POLEE_TRY(s.f[3].upload(ctx, y_grad, len));
    POLEE_TRY(s.f[4].upload(ctx, a_grad, len));
    POLEE_TRY(s.f[5].upload(ctx, b_grad, len));
    hipLaunchKernelGGL(kumaraswamy_grad_kernel, dim3(blocks(len)), dim3(256), 0, ctx->stream, s.f[0].p, s.f[1].p,
                       s.f[2].p, s.f[3].p, len, s.f[4].p, s.f[5].p);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(s.f[4].download(ctx, a_grad, len));
    return s.f[5].download(ctx, b_grad, len);
}

polee_status polee_gene_noninformative_prior(polee_ctx *ctx, const float *efflens, const float *xls, const float *xs,
                                             int32_t K, int64_t n, const int32_t *gene_of, double *x_grad)
{
    POLEE_TRY(use_device(ctx));
    if (!efflens || !xls || !xs || !gene_of || !x_grad || K < 1 || n < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    int32_t num_genes = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (gene_of[i] < -1) return fail(ctx, POLEE_ERR_BAD_ARG, "gene_of[%lld] = %d", (long long)i, gene_of[i]);
        num_genes = std::max(num_genes, gene_of[i] + 1);
    }
    if (num_genes == 0) return POLEE_OK;  // no gene information: nothing to add (likelihood-approximation.jl:489-492)
    const size_t tot = (size_t)n * K;
    DevBuf<float> d_l, d_xls, d_xs;
    DevBuf<int32_t> d_gene;
    DevBuf<int> d_k;
    DevBuf<double> d_c, d_sums, d_g;
    POLEE_TRY(d_l.upload(ctx, efflens, n));
    POLEE_TRY(d_xls.upload(ctx, xls, tot));
    POLEE_TRY(d_xs.upload(ctx, xs, tot));
    POLEE_TRY(d_gene.upload(ctx, gene_of, n));
    POLEE_TRY(d_g.upload(ctx, x_grad, tot));
    POLEE_TRY(d_c.alloc(ctx, (size_t)num_genes * K));
    POLEE_TRY(d_k.alloc(ctx, num_genes));
    POLEE_TRY(d_sums.alloc(ctx, 2 * (size_t)K));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d_c.p, 0, sizeof(double) * num_genes * K, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d_k.p, 0, sizeof(int) * num_genes, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d_sums.p, 0, sizeof(double) * 2 * K, ctx->stream));
    const dim3 grid(blocks(n), (unsigned)K);
    hipLaunchKernelGGL(gene_prior_sums_kernel, grid, dim3(256), 0, ctx->stream, d_l.p, d_xls.p, d_xs.p, d_gene.p, n, K,
                       d_c.p, d_k.p, d_sums.p);
    hipLaunchKernelGGL(gene_prior_offdiag_kernel, grid, dim3(256), 0, ctx->stream, d_xls.p, d_gene.p, n, K, d_c.p, d_k.p,
                       d_sums.p + K);
    hipLaunchKernelGGL(gene_prior_apply_kernel, grid, dim3(256), 0, ctx->stream, d_l.p, d_gene.p, n, K, d_c.p, d_k.p,
                       d_sums.p, d_sums.p + K, d_g.p);
    POLEE_KERNEL_CHECK(ctx);
    return d_g.download(ctx, x_grad, tot);
}

// test hook (include/polee_hip_debug.h): the forward kernel's double-precision exp, element-wise
polee_status polee_debug_fast_exp(polee_ctx *ctx, const double *x, int64_t count, double *out)
{
    POLEE_TRY(use_device(ctx));
    if (!x || !out || count < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    DevBuf<double> d;
    POLEE_TRY(d.upload(ctx, x, (size_t)count));
    hipLaunchKernelGGL(fast_exp_kernel, dim3(blocks(count)), dim3(256), 0, ctx->stream, d.p, count);
    POLEE_KERNEL_CHECK(ctx);
    return d.download(ctx, out, (size_t)count);
}

// test hook (include/polee_hip_debug.h): the tree kernels' double-precision log, element-wise
polee_status polee_debug_fast_log(polee_ctx *ctx, const double *x, int64_t count, double *out)
{
    POLEE_TRY(use_device(ctx));
    if (!x || !out || count < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    DevBuf<double> d;
    POLEE_TRY(d.upload(ctx, x, (size_t)count));
    hipLaunchKernelGGL(fast_log_kernel, dim3(blocks(count)), dim3(256), 0, ctx->stream, d.p, count);
    POLEE_KERNEL_CHECK(ctx);
    return d.download(ctx, out, (size_t)count);
}

}  // extern "C"
