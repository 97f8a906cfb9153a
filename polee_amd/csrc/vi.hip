// The likelihood-approximation VI loop, device resident.
// Replaces approximate_likelihood(::LogitSkewNormalPTTApprox, sample)
// (src/likelihood-approximation.jl:395-575), ADAM (:107-146), the factored variant
// (:248-392), and the element-wise reparameterisations (src/logitnormal.jl:8-55,
// src/sinh_arcsinh.jl:10-38), plus the sampler (src/approx-sampler.jl:37-44).
//
// One VI iteration = the K Monte-Carlo draws evaluated TOGETHER (one pass over X):
//   sample   : z0 -> sinh-arcsinh -> logit-normal -> clamp              -> ys [K][n-1] f64
//   forward  : Euler-tour scan (ptt_forward_device)                      -> xs [n][K] f32, leaf u
//   loglik   : loglik_eval_device (PSELL kernel)                         -> g  [n][K] f32
//   backward : double-double scan of u*(g - efflen term) over leaf order -> C  [K][n+1]
//   update   : per internal node: y_grad from C, chain rule through both
//              reparameterisations, mean over K, finiteness flag, ADAM   -> mu, omega, alpha
// No host synchronisation inside polee_vi_run.
#include "loglik_internal.hpp"
#include "ptt_internal.hpp"
#include "vi_fused.hpp"
#include "comm_internal.hpp"
#include "rng.hpp"

#include <cmath>

namespace polee {

struct NoiseSrc {
    const float *z0;  // device [steps][K][n-1] or null
    uint64_t seed;
    int32_t K;
    int64_t nm1;
    __device__ inline float get(int step /*1-based*/, int d, int64_t k) const
    {
        if (z0) return z0[((int64_t)(step - 1) * K + d) * nm1 + k];
        return philox_randn(seed, (uint32_t)step, (uint32_t)d, (uint32_t)k);
    }
    // all KK draws of node k at once (one Philox block per four draws)
    template <int KK>
    __device__ inline void get_all(int step, int64_t k, float (&z)[KK]) const
    {
        if (z0) {
#pragma unroll
            for (int d = 0; d < KK; ++d) z[d] = z0[((int64_t)(step - 1) * K + d) * nm1 + k];
            return;
        }
#pragma unroll
        for (int g = 0; g < (KK + 3) / 4; ++g) {
            float q[4];
            philox_randn4(seed, (uint32_t)step, (uint32_t)g, (uint32_t)k, q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < KK) z[4 * g + e] = q[e];
        }
    }
};

__device__ inline float logistic_f32(float x) { return 1.0f / (1.0f + expf(-x)); }  // logitnormal.jl:2

// elbo bookkeeping of the !gradonly mode: `elbo = lp + skew_ladj + ln_ladj + hsb_ladj` is an
// ASSIGNMENT inside the draw loop (likelihood-approximation.jl:537) followed by `/= K`
// (:561), i.e. the last draw's value over K.  lp_mean is the mean log-likelihood over draws.
__global__ void vi_trace_kernel(const double *lp, const double *ladj_el, const double *row_sums, int K, int idx,
                                double *elbo_trace, double *lp_trace)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int d = K - 1;
    elbo_trace[idx] = (lp[d] + ladj_el[d * 2 + 0] + ladj_el[d * 2 + 1] + row_sums[d * 2 + 1]) / K;
    double s = 0.0;
    for (int i = 0; i < K; ++i) s += lp[i];
    lp_trace[idx] = s / K;
}

__global__ void vi_init_mu_kernel(const double *ys, int64_t nm1, float *mu, float *omega, float *alpha)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nm1) return;
    const double y = ys[k];
    mu[k] = (float)log(y / (1 - y));  // logit (logitnormal.jl:4)
    omega[k] = logf(0.1f);
    alpha[k] = 0.0f;
}

// (index_of: leaf-order mode -- transcript j's values live at index_of[j])
__global__ void aos_to_rows_f32_kernel(const float *in, int K, int64_t n, float *out, const uint32_t *index_of)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int k = (int)(i / n);
    const int64_t j = i - (int64_t)k * n;
    out[i] = in[(index_of ? (int64_t)index_of[j] : j) * K + k];
}

// dst[i] = map[src[i]] / dst[i] = src[idx[i]]: the sparse pass's tables translated into the fit's leaf-order numbering
__global__ void remap_u32_kernel(const uint32_t *src, const uint32_t *map, int64_t count, uint32_t *dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = map[src[i]];
}
template <typename T>
__global__ void gather_kernel(const T *src, const int32_t *idx, int64_t count, T *dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[idx[i]];
}

// x_grad after the effective length adjustment, as rows [K][n] f64 (test hook output).
__global__ void vi_xgrad_rows_kernel(const float *g, const float *efflens, const double *csum, GenePrior gp, int K,
                                     int64_t n, double *out, const uint32_t *index_of)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int k = (int)(i / n);
    // (leaf-order mode: g, efflens and gene_of are indexed by leaf position; the output stays in transcript order)
    const int64_t tid = index_of ? (int64_t)index_of[i - (int64_t)k * n] : i - (int64_t)k * n;
    double xg = (double)g[tid * K + k];
    if (efflens) xg -= (double)((float)n * (1.0f / efflens[tid])) / csum[k];
    if (gp.gene_of) {  // (as bwd_values)
        const int gene = gp.gene_of[tid];
        const int kg = gene >= 0 ? gp.gene_k[gene] : 0;
        const double c = csum[k], inv_l = (double)(1.0f / efflens[tid]);
        const double xlg = kg > 1 ? -(double)(kg - 1) / gp.gene_c[(size_t)gene * K + k] : 0.0;
        xg += xlg * (inv_l / c) + inv_l * (gp.M / (c * c));
    }
    out[i] = xg;
}

// inverse_transform!(t, fill(1.0f0/n, n), ys) (likelihood-approximation.jl:451): every leaf holds 1/n
struct UniformLeafLoad {
    double val;
    __device__ dd operator()(int, int64_t) const { return dd_make(val); }
};
__global__ void vi_inverse_nodes_kernel(PttView v, const dd *C, double *ys)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= v.n - 1) return;
    const int lo = v.lo[k], mid = v.mid[k], hi1 = v.hi1[k];
    const double ur = dd_diff(C[mid], C[lo]);
    const double ul = dd_diff(C[hi1], C[mid]);
    ys[k] = ul / (ul + ur);
}

// sampler: rand! (approx-sampler.jl:37-44) -- no clamp of ys
// (y_eps > 0: the clamp of the initial-value draws, estimate.jl:443-447)
__global__ void sampler_y_kernel(const float *mu, const float *sigma, const float *alpha, NoiseSrc noise, double y_eps,
                                 double *ys)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (k >= noise.nm1) return;
    const float z0 = noise.get(1, d, k);
    const float zs = sinhf(alpha[k] + asinhf(z0));
    double y = (double)logistic_f32(mu[k] + zs * sigma[k]);
    if (y_eps > 0.0) y = y < y_eps ? y_eps : (y > 1 - y_eps ? 1 - y_eps : y);
    ys[(int64_t)d * noise.nm1 + k] = y;
}

__global__ void export_noise_kernel(NoiseSrc noise, int step, float *out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (k < noise.nm1) out[(int64_t)d * noise.nm1 + k] = noise.get(step, d, k);
}

}  // namespace polee

using namespace polee;

struct polee_vi {
    polee_ctx *ctx = nullptr;
    polee_loglik *ll = nullptr;
    polee_ptt *t = nullptr;
    polee_vi_opts o;
    int32_t n = 0, K = 0;
    int32_t step = 0;  // steps completed
    int32_t ahead_step = 0;  // iteration whose draws (ys, lyy) are already on the device; 0 = none
    int32_t trace_cap = 0;
    polee_comm *comm = nullptr;  // row-sharded fit: sum the likelihood gradient (and lp) over ranks each pass
    DevBuf<float> d_efflens, d_mu, d_omega, d_alpha, d_mm, d_vm, d_mo, d_vo, d_ma, d_va, d_z0, d_x, d_g;
    DevBuf<float> d_zcur;  // [n-1][K] the current iteration's N(0,1) draws
    DevBuf<float> d_y32;   // [n-1][K] and the logistic values they give (sample_node)
    DevBuf<float> d_u32;   // [n][K] leaf u of the forward pass (leaf order), unclamped, for the backward pass
    DevBuf<uint32_t> d_open_ptr, d_open_code;  // per forward-scan chunk: the tour's ENTER entries still open at its start
    // Leaf-order mode (round 4): this fit numbers the transcripts by their position among the tree's leaves.  d_x, d_g,
    // d_efflens and d_gene_of are indexed by leaf position -- the tree kernels' accesses to them are contiguous where
    // they were gathers through leaf_tid -- and the sparse pass gets its transcript-naming tables translated once, here
    // (LoglikRemap).  Results leave the handle in transcript order (index_of).
    bool leaf_order = false;
    DevBuf<uint32_t> d_index_of, d_dict_leaf, d_csr_col_leaf;
    DevBuf<float> d_single_leaf;
    LoglikRemap remap{nullptr, nullptr, nullptr, nullptr, false};
    DevBuf<double> d_ys, d_lyy, d_uleaf, d_part_c, d_part_ladj, d_csum, d_lp, d_ladj_el, d_rows, d_elbo, d_lptrace;
    DevBuf<dd> d_C;
    // backward (vi_bwd_local_kernel): the chunks' totals / offsets, the first node of every chunk's run, the prefix rows that
    // chunk-crossing nodes read (bit per leaf position), the subtree sums of the nodes inside one chunk [2][K][n-1]
    DevBuf<dd> d_chunk_bu;
    DevBuf<int32_t> d_node_start;
    DevBuf<uint32_t> d_need;
    DevBuf<float> d_H;
    int32_t bu_ch = 0;
    DevBuf<int> d_flag;
    // gene_noninformative (opts.gene_of): gene of every transcript, members per gene, per-gene sums of a step
    DevBuf<int32_t> d_gene_of;
    DevBuf<int> d_gene_k;
    DevBuf<double> d_gene_c;
    int32_t num_genes = 0;
    double gene_M = 0.0;
    // outputs of the test hook
    DevBuf<double> d_ygrad, d_xgrad_rows;
    DevBuf<float> d_mug, d_omg, d_alg, d_x_rows;

    NoiseSrc noise() const { return NoiseSrc{d_z0.p, o.seed, K, (int64_t)n - 1}; }
    polee_status one_step(bool apply, bool want_values, bool hook_outputs);
};

template <int K>
static polee_status vi_step_k(polee_vi *vi, bool apply, bool want_values, bool hook_outputs)
{
    polee_ctx *ctx = vi->ctx;
    polee_ptt *t = vi->t;
    const polee_vi_opts &o = vi->o;
    const int32_t n = vi->n;
    const int64_t nm1 = n - 1;
    const int step_num = vi->step + 1;
    hipStream_t st = ctx->stream;
    const PttView view = t->view();
    PttView lview = view;  // the view of the kernels that touch x / g: in leaf-order mode without the leaf -> transcript table
    if (vi->leaf_order) lview.leaf_tid = nullptr;
    const LoglikRemap *remap = vi->leaf_order ? &vi->remap : nullptr;
    const uint32_t *index_of = vi->leaf_order ? vi->d_index_of.p : nullptr;
    const int nch_f = fwd_num_chunks(t->TL);
    const float *eff = o.use_efflen_jacobian ? vi->d_efflens.p : nullptr;
    VK<K> *chunk_f = reinterpret_cast<VK<K> *>(t->d_chunk.p);
    const NoiseSrc noise = vi->noise();

    // sample
    if (want_values) POLEE_HIP_TRY(ctx, hipMemsetAsync(vi->d_ladj_el.p, 0, sizeof(double) * K * 2, st));
    // (skipped when the previous update already drew this step's samples, see below)
    if (nm1 > 0 && (vi->ahead_step != step_num || want_values)) {
        hipLaunchKernelGGL((vi_sample_k_kernel<K, NoiseSrc>), dim3((unsigned)ceil_div(nm1, 256)), dim3(256), 0, st,
                           vi->d_mu.p, vi->d_omega.p, vi->d_alpha.p, noise, step_num, vi->d_y32.p,
                           vi->d_zcur.p, want_values ? vi->d_ladj_el.p : nullptr);
        POLEE_KERNEL_CHECK(ctx);
    }
    // forward: xs = clamp(transform!(ys)) (likelihood-approximation.jl:525-526); also zeroes g
    // (up to 2048 chunks every apply workgroup sums the totals of the chunks before it itself: no spine launch)
    const int own_f = nch_f <= 2048;
    const YRows ysrc{vi->d_y32.p, o.y_eps};
    const bool open_lists = vi->d_open_ptr.p != nullptr;
    // (Experiment, POLEE_VI_XWIN_FOLD=1: the forward kernel also writes the sparse pass's x windows through the slot lists,
    // so that the gather launch in front of the pass goes.  Measured at C2: the pass loses its 6.9 us gather, the forward
    // kernel gains ~17 us of scattered 24-byte stores behind two dependent loads -- 2 570 against 2 695 iterations/s.  Off.)
    static const bool xwin_fold = getenv("POLEE_VI_XWIN_FOLD") != nullptr && getenv("POLEE_NO_RING") == nullptr;
    const bool xwin_here = xwin_fold && !vi->leaf_order && !vi->ll->force_mixed && vi->ll->d_xwin.p && vi->ll->d_tslot_ptr.p && vi->ll->host.num_tiles_s > 0;
    if (open_lists) {
        // (the chunks' offsets come from the tree's open-edge lists: no reduce launch)
    } else if (nch_f > 1) {
        hipLaunchKernelGGL((vi_fwd_reduce_kernel<K, YRows>), dim3(nch_f), dim3(SCAN_THREADS), 0, st, view, ysrc, chunk_f);
        if (!own_f) hipLaunchKernelGGL((scan_spine_kernel<VK<K>>), dim3(1), dim3(SCAN_THREADS), 0, st, chunk_f, nch_f);
    } else {
        POLEE_HIP_TRY(ctx, hipMemsetAsync(chunk_f, 0, sizeof(VK<K>), st));
    }
    if (want_values) {
        hipLaunchKernelGGL((vi_fwd_apply_kernel<K, YRows, float, true>), dim3(nch_f), dim3(SCAN_THREADS), 0, st, lview, ysrc, chunk_f,
                           vi->d_u32.p, vi->d_x.p, vi->d_g.p, eff, (float)o.y_eps, (float)(1.0 - o.y_eps),
                           eff ? vi->d_part_c.p : nullptr, want_values ? vi->d_part_ladj.p : nullptr, nch_f > 1 ? own_f : 0,
                           (const uint32_t *)vi->d_open_ptr.p, (const uint32_t *)vi->d_open_code.p,
                           xwin_here ? (const uint32_t *)vi->ll->d_tslot_ptr.p : nullptr, xwin_here ? (const uint32_t *)vi->ll->d_tslot.p : nullptr,
                           xwin_here ? vi->ll->d_xwin.p : nullptr,
                           remap && remap->singles_in_g ? (const float *)vi->d_single_leaf.p : nullptr);
    } else {
        hipLaunchKernelGGL((vi_fwd_apply_kernel<K, YRows, float, false>), dim3(nch_f), dim3(SCAN_THREADS), 0, st, lview, ysrc, chunk_f,
                           vi->d_u32.p, vi->d_x.p, vi->d_g.p, eff, (float)o.y_eps, (float)(1.0 - o.y_eps),
                           eff ? vi->d_part_c.p : nullptr, want_values ? vi->d_part_ladj.p : nullptr, nch_f > 1 ? own_f : 0,
                           (const uint32_t *)vi->d_open_ptr.p, (const uint32_t *)vi->d_open_code.p,
                           xwin_here ? (const uint32_t *)vi->ll->d_tslot_ptr.p : nullptr, xwin_here ? (const uint32_t *)vi->ll->d_tslot.p : nullptr,
                           xwin_here ? vi->ll->d_xwin.p : nullptr,
                           remap && remap->singles_in_g ? (const float *)vi->d_single_leaf.p : nullptr);
    }
    POLEE_KERNEL_CHECK(ctx);
    // likelihood
    if (want_values) POLEE_HIP_TRY(ctx, hipMemsetAsync(vi->d_lp.p, 0, sizeof(double) * PSELL_MAX_K, st));
    {
        const bool det_saved = vi->ll->deterministic;
        // (0 = by default exactly when the sample is shared by more than one rank; -1 = never)
        if (o.deterministic > 0 || (o.deterministic == 0 && vi->comm && vi->comm->nranks > 1)) vi->ll->deterministic = true;
        // (sum x / efflen over the forward kernel's per-chunk partials rides along with the pass's x-window gather)
        vi->ll->side_part = eff ? vi->d_part_c.p : nullptr;
        vi->ll->side_nparts = nch_f;
        vi->ll->side_out = vi->d_csum.p;
        vi->ll->side_done = false;
        const polee_status ls = loglik_eval_device(vi->ll, vi->d_x.p, K, vi->d_g.p, want_values ? vi->d_lp.p : nullptr, xwin_here, remap);
        vi->ll->deterministic = det_saved;
        const bool side_done = vi->ll->side_done;
        vi->ll->side_part = nullptr;
        POLEE_TRY(ls);
        if (eff && !side_done)  // (a pass without the gather launch)
            hipLaunchKernelGGL((vi_csum_finish_kernel<K>), dim3(1), dim3(256), 0, st, (const double *)vi->d_part_c.p, nch_f, vi->d_csum.p);
    }
    if (vi->comm) {  // this rank saw only its block of fragments: x_grad and lp are sums over fragments
        POLEE_TRY(comm_allreduce_device(vi->comm, vi->d_g.p, (size_t)n * K, false));
        if (want_values) POLEE_TRY(comm_allreduce_device(vi->comm, vi->d_lp.p, (size_t)K, true));
    }
    // gene_noninformative: the per-gene sums of this step's draws (likelihood-approximation.jl:535-538)
    GenePrior gp{nullptr, nullptr, nullptr, 0.0};
    if (vi->num_genes > 0) {
        gp = GenePrior{vi->d_gene_of.p, vi->d_gene_k.p, vi->d_gene_c.p, vi->gene_M};
        POLEE_HIP_TRY(ctx, hipMemsetAsync(vi->d_gene_c.p, 0, sizeof(double) * (size_t)vi->num_genes * K, st));
        hipLaunchKernelGGL((vi_gene_sums_kernel<K>), dim3((unsigned)ceil_div((int64_t)n, 256)), dim3(256), 0, st, vi->d_x.p,
                           vi->d_efflens.p, vi->d_part_c.p, nch_f, (int64_t)n, gp, vi->d_gene_c.p);
    }
    // backward + update: the chunk-local double-double prefix of u * (g - efflen term), the chunks' offsets, the update
    AdamConsts a;
    // adam_learning_rate(step_num - 1) (likelihood-approximation.jl:107-110, 497)
    a.lr = std::max(o.adam_min_learning_rate,
                    o.adam_initial_learning_rate * std::exp(-o.adam_learning_rate_decay * (double)(step_num - 1)));
    a.rm = o.adam_rm;
    a.rv = o.adam_rv;
    a.eps = o.adam_eps;
    a.m_denom = 1 - std::pow(o.adam_rm, (double)step_num);
    a.v_denom = 1 - std::pow(o.adam_rv, (double)step_num);
    a.inv_m_denom = 1.0 / a.m_denom;
    a.inv_v_denom = 1.0 / a.v_denom;
    a.max_mu = o.max_mu_step;
    a.max_omega = o.max_omega_step;
    a.max_alpha = o.max_alpha_step;
    a.first = step_num == 1;
    // the update also draws the next iteration's samples (one launch and one pass over the parameters less),
    // unless the caller's noise table ends here
    const bool sample_next = apply && !(o.z0 && step_num + 1 > o.num_steps);
    const UpdArgs ua{vi->d_y32.p, vi->d_mu.p, vi->d_omega.p, vi->d_alpha.p, vi->d_mm.p, vi->d_vm.p, vi->d_mo.p, vi->d_vo.p,
                     vi->d_ma.p, vi->d_va.p, a, apply ? 1 : 0, vi->d_flag.p, hook_outputs ? vi->d_ygrad.p : nullptr,
                     hook_outputs ? vi->d_mug.p : nullptr, hook_outputs ? vi->d_omg.p : nullptr,
                     hook_outputs ? vi->d_alg.p : nullptr, sample_next ? 1 : 0, o.y_eps, vi->d_zcur.p, step_num};
    {
        const int nch_bu = (int)ceil_div((int64_t)n, bu_ch<K>());
        VD<K> *chunk_bu = reinterpret_cast<VD<K> *>(vi->d_chunk_bu.p);
        const BwdArgs<K> ba{lview, vi->d_u32.p, vi->d_g.p, eff, vi->d_csum.p, gp, chunk_bu, vi->d_C.p, (const uint32_t *)vi->d_need.p,
                            (const int32_t *)vi->d_node_start.p, vi->d_H.p};
        hipLaunchKernelGGL((vi_bwd_local_kernel<K>), dim3(nch_bu), dim3(256), 0, st, ba);
        if (nm1 > 0) {
            hipLaunchKernelGGL((vi_bwd_spine_kernel<K>), dim3(1), dim3(64 * K), 0, st, chunk_bu, nch_bu);
            hipLaunchKernelGGL((vi_update_k_kernel<K, NoiseSrc>), dim3((unsigned)ceil_div(nm1, 256)), dim3(256), 0, st, view,
                               (const float *)vi->d_H.p, (const VD<K> *)chunk_bu, (const dd *)vi->d_C.p, ua, noise);
        }
        if (nm1 > 0) vi->ahead_step = apply ? (sample_next ? step_num + 1 : 0) : step_num;
    }
    POLEE_KERNEL_CHECK(ctx);
    if (want_values) {
        hipLaunchKernelGGL((vi_values_finish_kernel<K>), dim3(1), dim3(256), 0, st, vi->d_part_ladj.p, nch_f,
                           eff ? vi->d_csum.p : nullptr, vi->d_rows.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    if (hook_outputs) {
        hipLaunchKernelGGL(vi_xgrad_rows_kernel, dim3((unsigned)ceil_div((int64_t)n * K, 256)), dim3(256), 0, st,
                           vi->d_g.p, eff, vi->d_csum.p, gp, K, (int64_t)n, vi->d_xgrad_rows.p, index_of);
        POLEE_KERNEL_CHECK(ctx);
    }
    if (want_values && apply && vi->step < vi->trace_cap) {
        hipLaunchKernelGGL(vi_trace_kernel, dim3(1), dim3(64), 0, st, vi->d_lp.p, vi->d_ladj_el.p, vi->d_rows.p, K,
                           vi->step, vi->d_elbo.p, vi->d_lptrace.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    if (apply) ++vi->step;
    return POLEE_OK;
}

polee_status polee_vi::one_step(bool apply, bool want_values, bool hook_outputs)
{
    POLEE_TRY(t->reserve(K));
    if (o.z0 && step + 1 > o.num_steps)
        return fail(ctx, POLEE_ERR_BAD_ARG, "caller-supplied z0 covers only %d steps", o.num_steps);
    switch (K) {
        case 1: return vi_step_k<1>(this, apply, want_values, hook_outputs);
        case 2: return vi_step_k<2>(this, apply, want_values, hook_outputs);
        case 3: return vi_step_k<3>(this, apply, want_values, hook_outputs);
        case 4: return vi_step_k<4>(this, apply, want_values, hook_outputs);
        case 5: return vi_step_k<5>(this, apply, want_values, hook_outputs);
        case 6: return vi_step_k<6>(this, apply, want_values, hook_outputs);
        case 7: return vi_step_k<7>(this, apply, want_values, hook_outputs);
        case 8: return vi_step_k<8>(this, apply, want_values, hook_outputs);
    }
    return fail(ctx, POLEE_ERR_BAD_ARG, "num_mc_samples must be in 1..8");
}

namespace polee {

// ndraws draws of the sampler into d_all [ndraws][n] (device), in batches of 8 rows
static polee_status sampler_draw_device(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                        const float *z0, int32_t ndraws, uint64_t seed, DevBuf<float> &d_all,
                                        double y_eps = 0.0)
{
    polee_ctx *ctx = t->ctx;
    if (!mu || !sigma || !alpha || ndraws < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (t->T != 1) return fail(ctx, POLEE_ERR_BAD_ARG, "the sampler needs a single tree");
    const size_t n = t->n, nm1 = n - 1;
    DevBuf<float> d_mu, d_sigma, d_alpha, d_z0;
    POLEE_TRY(d_mu.upload(ctx, mu, nm1));
    POLEE_TRY(d_sigma.upload(ctx, sigma, nm1));
    POLEE_TRY(d_alpha.upload(ctx, alpha, nm1));
    POLEE_TRY(d_all.alloc(ctx, (size_t)ndraws * n));
    for (int32_t b0 = 0; b0 < ndraws; b0 += 8) {
        const int32_t B = std::min(8, ndraws - b0);
        POLEE_TRY(t->reserve(B));
        if (z0) POLEE_TRY(d_z0.upload(ctx, z0 + (size_t)b0 * nm1, (size_t)B * nm1));
        NoiseSrc noise{z0 ? d_z0.p : nullptr, seed + (uint64_t)b0 * 0x9E3779B97F4A7C15ull, B, (int64_t)nm1};
        if (nm1 > 0) {
            hipLaunchKernelGGL(sampler_y_kernel, dim3((unsigned)ceil_div(nm1, 256), B), dim3(256), 0, ctx->stream,
                               d_mu.p, d_sigma.p, d_alpha.p, noise, y_eps, t->d_ys.p);
            POLEE_KERNEL_CHECK(ctx);
        }
        FwdOut o;
        o.xs = d_all.p + (size_t)b0 * n;
        o.xs_rs = n;
        POLEE_TRY(ptt_forward_device(t, t->d_ys.p, B, o));
    }
    return POLEE_OK;
}

// posterior_mean (src/approx-sampler.jl:86-117): clamp every draw to [1e-15, 0.9999999], add them in draw order in
// f32 (pm .+= xs), divide by N
__global__ void sampler_mean_kernel(const float *xs, int ndraws, int64_t n, float *pm)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float acc = 0.0f;
    for (int k = 0; k < ndraws; ++k) acc += fminf(fmaxf(xs[(int64_t)k * n + j], 1e-15f), 0.9999999f);
    pm[j] = acc / (float)ndraws;
}

// initial values of the model entry (src/estimate.jl:436-455): every draw is divided by the effective lengths and
// renormalised (x0 ./= efflen; x0 ./= sum(x0), Float32), the draws are averaged
__global__ void x0_sums_kernel(const float *xs, const float *efflens, int64_t n, double *sums)
{
    __shared__ double smd[4];
    const int d = blockIdx.y;
    double q = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        q += (double)(xs[(int64_t)d * n + j] / efflens[j]);
    q = block_sum_f64(q, smd);
    if (threadIdx.x == 0) atomicAdd(&sums[d], q);
}
__global__ void x0_mean_kernel(const float *xs, const float *efflens, const double *sums, int ndraws, int64_t n, float *x0)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float acc = 0.0f;
    for (int d = 0; d < ndraws; ++d) acc += (xs[(int64_t)d * n + j] / efflens[j]) / (float)sums[d];
    x0[j] = acc / (float)ndraws;
}

constexpr int SAMPLER_MAX_Q = 8;
struct QuantileSpec {
    int nq;
    int j[SAMPLER_MAX_Q];         // 1-based lower order statistic
    double gamma[SAMPLER_MAX_Q];  // interpolation weight
};
// Statistics.quantile per transcript over the draws (src/approx-sampler.jl:50-83): thread per transcript; the rank of
// every draw by counting (ties broken by draw index), no sort.  The column is staged in LDS when it fits.
__global__ __launch_bounds__(64) void sampler_quantile_kernel(const float *xs, int ndraws, int64_t n, QuantileSpec spec,
                                                              int in_lds, float *out)
{
    extern __shared__ float col[];  // [ndraws][64]
    const int64_t jj = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t j = jj < n ? jj : n - 1;
    if (in_lds)
        for (int k = 0; k < ndraws; ++k) col[k * 64 + threadIdx.x] = xs[(int64_t)k * n + j];
    auto at = [&](int k) { return in_lds ? col[k * 64 + threadIdx.x] : xs[(int64_t)k * n + j]; };
    float a[SAMPLER_MAX_Q], b[SAMPLER_MAX_Q];
#pragma unroll
    for (int i = 0; i < SAMPLER_MAX_Q; ++i) a[i] = b[i] = 0.0f;
    for (int e = 0; e < ndraws; ++e) {
        const float ve = at(e);
        int rank = 0;
        for (int i = 0; i < ndraws; ++i) {
            const float vi = at(i);
            rank += (vi < ve || (vi == ve && i < e)) ? 1 : 0;
        }
#pragma unroll
        for (int i = 0; i < SAMPLER_MAX_Q; ++i)
            if (i < spec.nq) {
                if (rank == spec.j[i] - 1) a[i] = ve;
                if (rank == spec.j[i] || ndraws == 1) b[i] = ve;
            }
    }
    if (jj < n)
#pragma unroll
        for (int i = 0; i < SAMPLER_MAX_Q; ++i)
            if (i < spec.nq) out[(int64_t)i * n + j] = (float)((double)a[i] + spec.gamma[i] * (double)(b[i] - a[i]));
}

}  // namespace polee
using namespace polee;

extern "C" {

void polee_vi_default_opts(polee_vi_opts *o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->num_steps = 500;           // LIKAP_NUM_STEPS (constants.jl:64)
    o->num_mc_samples = 6;        // LIKAP_NUM_MC_SAMPLES (constants.jl:65)
    o->use_efflen_jacobian = 1;
    o->gradonly = 1;
    o->seed = 123456789ull;       // main.jl:126
    o->z0 = nullptr;
    o->y_eps = 1e-10;             // LIKAP_Y_EPS (constants.jl:48)
    o->adam_initial_learning_rate = 1.0;
    o->adam_learning_rate_decay = 2e-2;
    o->adam_min_learning_rate = 1e-3;
    o->adam_eps = 1e-8;
    o->adam_rv = 0.9;
    o->adam_rm = 0.7;
    o->max_mu_step = 2e-1;        // likelihood-approximation.jl:421-423
    o->max_omega_step = 2e-1;
    o->max_alpha_step = 2e-2;
    o->profile = 0;
    o->deterministic = 0;
    o->gene_of = nullptr;
}

// (polee_optimize_ptt drives the K = 1 kernels itself, in transcript order)
static thread_local bool g_vi_plain_order = false;

polee_status polee_vi_create(polee_loglik *ll, polee_ptt *t, const float *efflens, const polee_vi_opts *opts,
                             polee_vi **out)
{
    if (!ll || !t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = ll->ctx;
    POLEE_TRY(use_device(ctx));
    if (t->ctx != ctx) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood and tree belong to different contexts");
    if (!out || !efflens) return fail(ctx, POLEE_ERR_BAD_ARG, "null argument");
    if (t->T != 1) return fail(ctx, POLEE_ERR_BAD_ARG, "the VI loop needs a single tree");
    if ((int64_t)t->n != ll->n)
        return fail(ctx, POLEE_ERR_BAD_ARG, "tree has %d leaves but X has %lld transcripts", t->n, (long long)ll->n);
    polee_vi_opts o;
    if (opts)
        o = *opts;
    else
        polee_vi_default_opts(&o);
    if (o.num_mc_samples < 1 || o.num_mc_samples > PSELL_MAX_K)
        return fail(ctx, POLEE_ERR_BAD_ARG, "num_mc_samples must be in 1..8");
    if (o.num_steps < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "num_steps must be >= 0");
    // (the forward kernel takes log y and log(1 - y) of the clamped y without the special cases of a general log)
    if (!(o.y_eps > 0.0 && o.y_eps < 0.5)) return fail(ctx, POLEE_ERR_BAD_ARG, "y_eps must lie in (0, 0.5) (LIKAP_Y_EPS = 1e-10)");
    // gene_noninformative: members per gene and M = sum over genes of (members - 1)
    std::vector<int> gene_k;
    double gene_M = 0.0;
    if (o.gene_of) {
        if (!o.use_efflen_jacobian)
            return fail(ctx, POLEE_ERR_BAD_ARG, "gene_of needs use_efflen_jacobian (gene_noninformative_prior! works on the "
                                                "xls of the effective-length adjustment, likelihood.jl:114-159)");
        int32_t ng = 0;
        for (int64_t i = 0; i < ll->n; ++i) {
            if (o.gene_of[i] < -1) return fail(ctx, POLEE_ERR_BAD_ARG, "gene_of[%lld] = %d", (long long)i, o.gene_of[i]);
            ng = std::max(ng, o.gene_of[i] + 1);
        }
        gene_k.assign((size_t)ng, 0);
        for (int64_t i = 0; i < ll->n; ++i)
            if (o.gene_of[i] >= 0) ++gene_k[o.gene_of[i]];
        for (int k : gene_k)
            if (k > 1) gene_M += (double)(k - 1);
        // (no gene information at all: the reference warns and switches the option off, :487-490)
    }
    polee_vi *vi = new (std::nothrow) polee_vi();
    if (!vi) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    vi->ctx = ctx;
    vi->ll = ll;
    vi->t = t;
    ctx_retain(ctx);
    loglik_retain(ll);
    ptt_retain(t);
    vi->o = o;
    vi->n = t->n;
    vi->K = o.num_mc_samples;
    ll->profile = o.profile != 0;
    ll->prof_every = o.profile > 1 ? o.profile : 1;
    // (opts.deterministic belongs to THIS fit: applied around its likelihood passes, the handle's own setting restored)
    const size_t n = vi->n, nm1 = std::max<size_t>(n - 1, 1), K = vi->K;
    polee_status s = POLEE_OK;
    auto A = [&](polee_status r) {
        if (s == POLEE_OK) s = r;
    };
    A(vi->d_efflens.upload(ctx, efflens, n));
    for (DevBuf<float> *b : {&vi->d_mu, &vi->d_omega, &vi->d_alpha, &vi->d_mm, &vi->d_vm, &vi->d_mo, &vi->d_vo,
                             &vi->d_ma, &vi->d_va, &vi->d_mug, &vi->d_omg, &vi->d_alg})
        A(b->alloc(ctx, nm1));
    A(vi->d_x.alloc(ctx, n * K));
    A(vi->d_zcur.alloc(ctx, nm1 * K));
    A(vi->d_y32.alloc(ctx, nm1 * K));
    A(vi->d_g.alloc(ctx, n * K));
    A(vi->d_x_rows.alloc(ctx, n * K));
    A(vi->d_ys.alloc(ctx, nm1));  // (initial values; the point optimisation's y)
    if (g_vi_plain_order) A(vi->d_lyy.alloc(ctx, nm1 * 2));  // (the point optimisation's edge logs)
    A(vi->d_u32.alloc(ctx, n * K));
    if (g_vi_plain_order) A(vi->d_uleaf.alloc(ctx, n));  // (the point optimisation's f64 leaf u)
    A(vi->d_C.alloc(ctx, (n + 1) * K));
    {
        const size_t nch = (size_t)std::max(scan_num_chunks(3 * (int64_t)n - 2), 1);
        A(vi->d_part_c.alloc(ctx, nch * K));
        A(vi->d_part_ladj.alloc(ctx, nch * K));
        A(vi->d_csum.alloc(ctx, 2 * PSELL_MAX_K));
    }
    {   // backward tables (vi_bwd_local_kernel): internal nodes in DFS pre-order have non-decreasing lo
        const PttPlan &pl = t->plans[0];
        const int CH = K <= 6 ? bu_ch<6>() : bu_ch<8>();
        vi->bu_ch = CH;
        const int nch_bu = (int)ceil_div((int64_t)n, CH);
        std::vector<int32_t> node_start((size_t)nch_bu + 1, (int32_t)(n - 1));
        std::vector<uint32_t> need((n + 1 + 31) / 32 + 1, 0u);
        bool monotone = true;
        int c = 0;
        for (size_t k = 0; k + 1 < n; ++k) {
            if (k > 0 && pl.lo[k] < pl.lo[k - 1]) monotone = false;
            for (; c <= pl.lo[k] / CH; ++c) node_start[(size_t)c] = (int32_t)k;
            const int64_t end = std::min<int64_t>(((int64_t)pl.lo[k] / CH + 1) * CH, (int64_t)n);
            if (pl.hi1[k] > end)
                for (int32_t b : {pl.lo[k], pl.mid[k], pl.hi1[k]}) need[(size_t)b >> 5] |= 1u << (b & 31);
        }
        if (!monotone) A(fail(ctx, POLEE_ERR_BAD_ARG, "tree plan: internal nodes are not in DFS pre-order"));
        A(vi->d_node_start.upload(ctx, node_start));
        A(vi->d_need.upload(ctx, need));
        A(vi->d_chunk_bu.alloc(ctx, ((size_t)nch_bu + 1) * K));
        A(vi->d_H.alloc(ctx, 2 * K * nm1));
        // (row n of C is exported by the kernel when a node needs it, unless n is a multiple of the chunk size: then it is the
        // zero first row of a chunk past the end)
        if (s == POLEE_OK && hipMemsetAsync(vi->d_C.p, 0, sizeof(dd) * (n + 1) * K, ctx->stream) != hipSuccess)
            A(fail(ctx, POLEE_ERR_HIP, "memset failed"));
    }
    {   // The forward scan's chunk offsets from the tree (vi_fwd_apply_kernel): for every chunk of the Euler tour the ENTER
        // entries still open at its first entry = the path from the root to that point.  One walk over the tour with a
        // stack; kept when the lists stay small (a caterpillar tree of 200 000 leaves would need n^2 / 1024 entries: the
        // reduce launch stays for such trees).
        static const bool no_open = getenv("POLEE_VI_NO_OPEN_LISTS") != nullptr;  // (A/B)
        const std::vector<uint32_t> &code = t->plans[0].tour_code;
        const int64_t TL = t->TL;
        const size_t limit = (size_t)8 << 20;
        std::vector<uint32_t> optr, ocode;
        const bool ok = !no_open && t->T == 1 && (int64_t)code.size() == TL && build_open_lists(code.data(), TL, FWD_CHUNK, limit, optr, ocode);
        if (ok) {
            if (ocode.empty()) ocode.push_back(4u | TOUR_LEAF);  // (never read)
            A(vi->d_open_ptr.upload(ctx, optr));
            A(vi->d_open_code.upload(ctx, ocode));
        }
    }
    A(vi->d_ygrad.alloc(ctx, nm1 * K));
    A(vi->d_xgrad_rows.alloc(ctx, n * K));
    A(vi->d_lp.alloc(ctx, PSELL_MAX_K));
    A(vi->d_ladj_el.alloc(ctx, PSELL_MAX_K * 2));
    A(vi->d_rows.alloc(ctx, PSELL_MAX_K * 2));
    A(vi->d_flag.alloc(ctx, 1));
    if (!gene_k.empty()) {
        vi->num_genes = (int32_t)gene_k.size();
        vi->gene_M = gene_M;
        A(vi->d_gene_of.upload(ctx, o.gene_of, n));
        A(vi->d_gene_k.upload(ctx, gene_k));
        A(vi->d_gene_c.alloc(ctx, gene_k.size() * K));
    }
    // Leaf-order mode: the transcripts renumbered by leaf position for this fit (polee_vi::leaf_order)
    static const bool no_leaf_env = getenv("POLEE_VI_NO_LEAF_ORDER") != nullptr;  // (A/B)
    if (!no_leaf_env && !g_vi_plain_order && n > 1 && t->plans.size() == 1 && t->plans[0].tid_pos.size() == n) {
        const PttPlan &pl = t->plans[0];
        std::vector<uint32_t> index_of(n);
        std::vector<float> eff_leaf(n);
        for (size_t j = 0; j < n; ++j) index_of[j] = (uint32_t)pl.tid_pos[j];
        for (size_t pos = 0; pos < n; ++pos) eff_leaf[pos] = efflens[pl.leaf_tid[pos]];
        A(vi->d_index_of.upload(ctx, index_of));
        A(vi->d_efflens.upload(ctx, eff_leaf));
        if (!gene_k.empty()) {
            std::vector<int32_t> gene_leaf(n);
            for (size_t pos = 0; pos < n; ++pos) gene_leaf[pos] = o.gene_of[pl.leaf_tid[pos]];
            A(vi->d_gene_of.upload(ctx, gene_leaf));
        }
        A(vi->d_dict_leaf.alloc(ctx, (size_t)std::max<int64_t>(ll->dict_len, 1)));
        if (ll->csr_nnz > 0) A(vi->d_csr_col_leaf.alloc(ctx, (size_t)ll->csr_nnz));
        if (ll->has_singles) A(vi->d_single_leaf.alloc(ctx, n));
        if (s == POLEE_OK) {
            hipStream_t st0 = ctx->stream;
            if (ll->dict_len > 0)
                hipLaunchKernelGGL(remap_u32_kernel, dim3((unsigned)ceil_div(ll->dict_len, 256)), dim3(256), 0, st0,
                                   (const uint32_t *)ll->d_dict.p, (const uint32_t *)vi->d_index_of.p, ll->dict_len, vi->d_dict_leaf.p);
            if (ll->csr_nnz > 0)
                hipLaunchKernelGGL(remap_u32_kernel, dim3((unsigned)ceil_div(ll->csr_nnz, 256)), dim3(256), 0, st0,
                                   (const uint32_t *)ll->d_csr_col.p, (const uint32_t *)vi->d_index_of.p, ll->csr_nnz, vi->d_csr_col_leaf.p);
            if (ll->has_singles)
                hipLaunchKernelGGL(gather_kernel<float>, dim3((unsigned)ceil_div((int64_t)n, 256)), dim3(256), 0, st0,
                                   (const float *)ll->d_single_cnt.p, (const int32_t *)t->d_leaf_tid.p, (int64_t)n, vi->d_single_leaf.p);
            if (hipGetLastError() != hipSuccess) A(fail(ctx, POLEE_ERR_HIP, "leaf-order tables: kernel launch failed"));
            vi->leaf_order = true;
            // (the singles' cnt / x goes into g in the forward kernel: one launch and one pass over g less per iteration)
            vi->remap = LoglikRemap{vi->d_dict_leaf.p, ll->has_singles ? vi->d_single_leaf.p : nullptr,
                                    ll->csr_nnz > 0 ? vi->d_csr_col_leaf.p : nullptr, vi->d_index_of.p, ll->has_singles};
        }
    }
    vi->o.gene_of = nullptr;  // (host memory is borrowed for the call only)
    vi->trace_cap = o.gradonly ? 0 : std::max(o.num_steps, 1);
    if (!o.gradonly) {
        A(vi->d_elbo.alloc(ctx, vi->trace_cap));
        A(vi->d_lptrace.alloc(ctx, vi->trace_cap));
    }
    if (o.z0 && o.num_steps > 0) A(vi->d_z0.upload(ctx, o.z0, (size_t)o.num_steps * K * (n - 1)));
    vi->o.z0 = o.z0;  // only its null-ness is used from here on
    A(t->reserve((int32_t)K));
    if (s != POLEE_OK) {
        polee_vi_destroy(vi);
        return s;
    }
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemsetAsync(vi->d_flag.p, 0, sizeof(int), st);
    if (e == hipSuccess) e = hipMemsetAsync(vi->d_rows.p, 0, sizeof(double) * PSELL_MAX_K * 2, st);
    for (DevBuf<float> *b : {&vi->d_mm, &vi->d_vm, &vi->d_mo, &vi->d_vo, &vi->d_ma, &vi->d_va})
        if (e == hipSuccess) e = hipMemsetAsync(b->p, 0, sizeof(float) * nm1, st);
    if (e != hipSuccess) {
        polee_vi_destroy(vi);
        return fail(ctx, POLEE_ERR_HIP, "memset failed: %s", hipGetErrorString(e));
    }
    // initial values (likelihood-approximation.jl:451-456): mu = logit(inverse_transform!(fill(1/n)))
    if (n > 1) {
        UniformLeafLoad load{(double)(1.0f / (float)n)};
        LeafPrefixEmit emit{(int32_t)n, t->d_C.p};
        e = run_scan_partial<dd>(st, 1, n, t->d_chunk.p, nullptr, load, emit);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(vi_inverse_nodes_kernel, dim3((unsigned)ceil_div(n - 1, 256)), dim3(256), 0, st,
                               t->view(), t->d_C.p, vi->d_ys.p);
            hipLaunchKernelGGL(vi_init_mu_kernel, dim3((unsigned)ceil_div(n - 1, 256)), dim3(256), 0, st, vi->d_ys.p,
                               (int64_t)n - 1, vi->d_mu.p, vi->d_omega.p, vi->d_alpha.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            polee_vi_destroy(vi);
            return fail(ctx, POLEE_ERR_HIP, "initialisation failed: %s", hipGetErrorString(e));
        }
    }
    *out = vi;
    return POLEE_OK;
}

void polee_vi_destroy(polee_vi *vi)
{
    if (!vi) return;
    polee_ctx *ctx = vi->ctx;
    polee_loglik *ll = vi->ll;
    polee_ptt *t = vi->t;
    polee_comm *comm = vi->comm;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    delete vi;
    polee_comm_destroy(comm);  // drops this handle's reference
    loglik_release(ll);
    ptt_release(t);
    ctx_release(ctx);
}

polee_status polee_vi_set_comm(polee_vi *vi, polee_comm *comm)
{
    if (!vi) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    if (comm && comm->ctx != vi->ctx)
        return fail(vi->ctx, POLEE_ERR_BAD_ARG, "communicator and fit belong to different contexts");
    if (comm) ++comm->refs;
    polee_comm_destroy(vi->comm);
    vi->comm = comm;
    return POLEE_OK;
}

polee_status polee_vi_run(polee_vi *vi, int32_t nsteps)
{
    if (!vi) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    POLEE_TRY(use_device(vi->ctx));
    vi->ll->profile = vi->o.profile != 0;
    vi->ll->prof_every = vi->o.profile > 1 ? vi->o.profile : 1;
    for (int32_t i = 0; i < nsteps; ++i) POLEE_TRY(vi->one_step(true, !vi->o.gradonly, false));
    return POLEE_OK;
}

polee_status polee_vi_sync(polee_vi *vi)
{
    if (!vi) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = vi->ctx;
    POLEE_TRY(use_device(ctx));
    int flag = 0;
    POLEE_TRY(vi->d_flag.download(ctx, &flag, 1));
    if (vi->ll->profile) POLEE_TRY(vi->ll->profile_collect());
    if (flag != 0)
        return fail(ctx, POLEE_ERR_NONFINITE, "non-finite gradient at VI step %d (likelihood-approximation.jl:559)",
                    flag);
    return POLEE_OK;
}

polee_status polee_vi_get_params(polee_vi *vi, float *mu, float *omega, float *alpha)
{
    if (!vi) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = vi->ctx;
    POLEE_TRY(use_device(ctx));
    const size_t nm1 = vi->n - 1;
    if (mu) POLEE_TRY(vi->d_mu.download(ctx, mu, nm1));
    if (omega) POLEE_TRY(vi->d_omega.download(ctx, omega, nm1));
    if (alpha) POLEE_TRY(vi->d_alpha.download(ctx, alpha, nm1));
    return POLEE_OK;
}

polee_status polee_vi_set_params(polee_vi *vi, const float *mu, const float *omega, const float *alpha)
{
    if (!vi) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = vi->ctx;
    POLEE_TRY(use_device(ctx));
    const size_t nm1 = vi->n - 1;
    if (mu) POLEE_TRY(vi->d_mu.upload(ctx, mu, nm1));
    if (omega) POLEE_TRY(vi->d_omega.upload(ctx, omega, nm1));
    if (alpha) POLEE_TRY(vi->d_alpha.upload(ctx, alpha, nm1));
    vi->ahead_step = 0;  // draws made from the old parameters are stale
    return POLEE_OK;
}

polee_status polee_vi_get_stats(polee_vi *vi, polee_vi_stats *stats)
{
    if (!vi || !stats) return fail(nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = vi->ctx;
    POLEE_TRY(use_device(ctx));
    memset(stats, 0, sizeof(*stats));
    stats->steps_done = vi->step;
    int flag = 0;
    POLEE_TRY(vi->d_flag.download(ctx, &flag, 1));
    stats->nonfinite_step = flag;
    if (vi->ll->profile) POLEE_TRY(vi->ll->profile_collect());
    stats->loglik_kernel_launches = vi->ll->prof_launches;
    stats->loglik_kernel_ms_avg = vi->ll->prof_launches ? vi->ll->prof_ms_total / vi->ll->prof_launches : 0.0;
    stats->loglik_pass_ms_avg = vi->ll->prof_launches ? vi->ll->prof_pass_ms_total / vi->ll->prof_launches : 0.0;
    if (!vi->o.gradonly && vi->step > 0 && vi->step <= vi->trace_cap) {
        POLEE_HIP_TRY(ctx, hipMemcpyAsync(&stats->last_elbo, vi->d_elbo.p + (vi->step - 1), sizeof(double),
                                          hipMemcpyDeviceToHost, ctx->stream));
        POLEE_HIP_TRY(ctx, hipMemcpyAsync(&stats->last_lp_mean, vi->d_lptrace.p + (vi->step - 1), sizeof(double),
                                          hipMemcpyDeviceToHost, ctx->stream));
        POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return POLEE_OK;
}

polee_status polee_vi_get_trace(polee_vi *vi, double *elbo, double *lp_mean)
{
    if (!vi) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = vi->ctx;
    POLEE_TRY(use_device(ctx));
    if (vi->o.gradonly) return fail(ctx, POLEE_ERR_BAD_ARG, "no trace is recorded in gradonly mode");
    const size_t cnt = std::min(vi->step, vi->trace_cap);
    if (elbo) POLEE_TRY(vi->d_elbo.download(ctx, elbo, cnt));
    if (lp_mean) POLEE_TRY(vi->d_lptrace.download(ctx, lp_mean, cnt));
    return POLEE_OK;
}

#ifdef POLEE_VI_STAMPS
// diagnostic build only: the tree kernels' phase stamps of their last launches, [4][2048][8] (vi_fused.hpp)
polee_status polee_debug_vi_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(polee::g_vi_stamps), sizeof(unsigned long long) * 4 * 2048 * 8) == hipSuccess ? POLEE_OK : POLEE_ERR_HIP;
}
#endif

polee_status polee_vi_export_noise(polee_vi *vi, int32_t step, float *z0)
{
    if (!vi || !z0 || step < 1) return fail(nullptr, POLEE_ERR_BAD_ARG, "bad argument");
    polee_ctx *ctx = vi->ctx;
    POLEE_TRY(use_device(ctx));
    const int64_t nm1 = vi->n - 1;
    if (nm1 < 1) return POLEE_OK;
    if (vi->o.z0 && step > vi->o.num_steps) return fail(ctx, POLEE_ERR_BAD_ARG, "step beyond the supplied noise");
    DevBuf<float> tmp;
    POLEE_TRY(tmp.alloc(ctx, (size_t)vi->K * nm1));
    hipLaunchKernelGGL(export_noise_kernel, dim3((unsigned)ceil_div(nm1, 256), vi->K), dim3(256), 0, ctx->stream,
                       vi->noise(), step, tmp.p);
    POLEE_KERNEL_CHECK(ctx);
    return tmp.download(ctx, z0, (size_t)vi->K * nm1);
}

polee_status polee_vi_eval_gradients(polee_vi *vi, float *xs, double *x_grad, double *y_grad, float *mu_grad,
                                     float *omega_grad, float *alpha_grad, double *lp, double *ladj)
{
    if (!vi) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = vi->ctx;
    POLEE_TRY(use_device(ctx));
    const size_t n = vi->n, nm1 = n - 1, K = vi->K;
    POLEE_TRY(vi->one_step(false, true, true));
    if (xs) {
        hipLaunchKernelGGL(aos_to_rows_f32_kernel, dim3((unsigned)ceil_div(n * K, 256)), dim3(256), 0, ctx->stream,
                           vi->d_x.p, (int)K, (int64_t)n, vi->d_x_rows.p, vi->leaf_order ? (const uint32_t *)vi->d_index_of.p : nullptr);
        POLEE_KERNEL_CHECK(ctx);
        POLEE_TRY(vi->d_x_rows.download(ctx, xs, n * K));
    }
    if (x_grad) POLEE_TRY(vi->d_xgrad_rows.download(ctx, x_grad, n * K));
    if (y_grad) POLEE_TRY(vi->d_ygrad.download(ctx, y_grad, nm1 * K));
    if (mu_grad) POLEE_TRY(vi->d_mug.download(ctx, mu_grad, nm1));
    if (omega_grad) POLEE_TRY(vi->d_omg.download(ctx, omega_grad, nm1));
    if (alpha_grad) POLEE_TRY(vi->d_alg.download(ctx, alpha_grad, nm1));
    if (lp) POLEE_TRY(vi->d_lp.download(ctx, lp, K));
    if (ladj) {
        std::vector<double> el(K * 2), rs(K * 2);
        POLEE_TRY(vi->d_ladj_el.download(ctx, el.data(), K * 2));
        POLEE_TRY(vi->d_rows.download(ctx, rs.data(), K * 2));
        for (size_t d = 0; d < K; ++d) ladj[d] = el[d * 2] + el[d * 2 + 1] + rs[d * 2 + 1];
    }
    return POLEE_OK;
}

polee_status polee_vi_fit(polee_loglik *ll, polee_ptt *t, const float *efflens, const polee_vi_opts *opts, float *mu,
                          float *omega, float *alpha, polee_vi_stats *stats)
{
    polee_vi *vi = nullptr;
    POLEE_TRY(polee_vi_create(ll, t, efflens, opts, &vi));
    polee_status s = polee_vi_run(vi, vi->o.num_steps);
    if (s == POLEE_OK) s = polee_vi_sync(vi);
    if (s == POLEE_OK) s = polee_vi_get_params(vi, mu, omega, alpha);
    if (s == POLEE_OK && stats) s = polee_vi_get_stats(vi, stats);
    polee_vi_destroy(vi);
    return s;
}

polee_status polee_optimize_ptt(polee_loglik *ll, polee_ptt *t, const float *efflens, int32_t num_steps, float *xs,
                                float *zs_out)
{
    if (!ll || !t || !efflens || !xs || num_steps < 0) return fail(nullptr, POLEE_ERR_BAD_ARG, "bad argument");
    polee_vi_opts o;
    polee_vi_default_opts(&o);
    o.num_steps = num_steps;
    o.num_mc_samples = 1;
    polee_vi *vi = nullptr;
    g_vi_plain_order = true;
    const polee_status cs = polee_vi_create(ll, t, efflens, &o, &vi);  // initial z = logit(inverse_transform(1/n)) lands in d_mu
    g_vi_plain_order = false;
    POLEE_TRY(cs);
    polee_ctx *ctx = vi->ctx;
    hipStream_t st = ctx->stream;
    const int32_t n = vi->n;
    const int64_t nm1 = n - 1;
    const PttView view = t->view();
    const int nch_f = fwd_num_chunks(t->TL), nch_b = scan_num_chunks(n);
    VK<1> *chunk_f = reinterpret_cast<VK<1> *>(t->d_chunk.p);
    VD<1> *chunk_b = reinterpret_cast<VD<1> *>(t->d_chunk.p);
    polee_status rc = POLEE_OK;
    auto forward = [&]() -> polee_status {
        hipLaunchKernelGGL(point_sample_kernel, dim3((unsigned)ceil_div(nm1, 256)), dim3(256), 0, st, vi->d_mu.p, nm1,
                           vi->d_ys.p, vi->d_lyy.p);
        if (nch_f > 1) {
            hipLaunchKernelGGL((vi_fwd_reduce_kernel<1, LogRows>), dim3(nch_f), dim3(SCAN_THREADS), 0, st, view, LogRows{vi->d_lyy.p}, chunk_f);
            hipLaunchKernelGGL((scan_spine_kernel<VK<1>>), dim3(1), dim3(SCAN_THREADS), 0, st, chunk_f, nch_f);
        } else {
            POLEE_HIP_TRY(ctx, hipMemsetAsync(chunk_f, 0, sizeof(VK<1>), st));
        }
        hipLaunchKernelGGL((vi_fwd_apply_kernel<1, LogRows, double, false>), dim3(nch_f), dim3(SCAN_THREADS), 0, st, view, LogRows{vi->d_lyy.p}, chunk_f,
                           vi->d_uleaf.p, vi->d_x.p, vi->d_g.p, vi->d_efflens.p, (float)o.y_eps, (float)(1.0 - o.y_eps),
                           vi->d_part_c.p, (double *)nullptr, 0, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, (float *)nullptr, (const float *)nullptr);
        POLEE_KERNEL_CHECK(ctx);
        return POLEE_OK;
    };
    for (int step_num = 1; step_num <= num_steps && rc == POLEE_OK && nm1 > 0; ++step_num) {
        rc = forward();
        if (rc != POLEE_OK) break;
        rc = loglik_eval_device(ll, vi->d_x.p, 1, vi->d_g.p, nullptr);
        if (rc != POLEE_OK) break;
        const GenePrior no_gp{nullptr, nullptr, nullptr, 0.0};
        hipLaunchKernelGGL((vi_bwd_reduce_kernel<1>), dim3(nch_b), dim3(SCAN_THREADS), 0, st, view, vi->d_uleaf.p,
                           vi->d_g.p, vi->d_efflens.p, vi->d_part_c.p, nch_f, vi->d_csum.p, no_gp, chunk_b);
        hipLaunchKernelGGL((scan_spine_kernel<VD<1>>), dim3(1), dim3(SCAN_THREADS), 0, st, chunk_b, nch_b);
        hipLaunchKernelGGL((vi_bwd_apply_kernel<1>), dim3(nch_b), dim3(SCAN_THREADS), 0, st, view, vi->d_uleaf.p,
                           vi->d_g.p, vi->d_efflens.p, vi->d_csum.p, no_gp, chunk_b, vi->d_C.p, 0);
        AdamConsts a;
        a.lr = std::max(o.adam_min_learning_rate,
                        o.adam_initial_learning_rate * std::exp(-o.adam_learning_rate_decay * (double)(step_num - 1)));
        a.rm = o.adam_rm;
        a.rv = o.adam_rv;
        a.eps = o.adam_eps;
        a.m_denom = 1 - std::pow(o.adam_rm, (double)step_num);
        a.v_denom = 1 - std::pow(o.adam_rv, (double)step_num);
        a.inv_m_denom = 1.0 / a.m_denom;
        a.inv_v_denom = 1.0 / a.v_denom;
        a.max_mu = 1e-1;  // ss_max_z_step (likelihood-approximation.jl:166)
        a.max_omega = a.max_alpha = 0.0;
        a.first = step_num == 1;
        hipLaunchKernelGGL(point_update_kernel, dim3((unsigned)ceil_div(nm1, 256)), dim3(256), 0, st, view, vi->d_ys.p,
                           vi->d_C.p, vi->d_mu.p, vi->d_mm.p, vi->d_vm.p, a, vi->d_flag.p, step_num);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(ctx, POLEE_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    }
    if (rc == POLEE_OK) rc = forward();  // final xs = clamp(transform(logistic(zs))) (:235-241)
    if (rc == POLEE_OK) rc = polee_vi_sync(vi);
    if (rc == POLEE_OK) rc = vi->d_x.download(ctx, xs, n);  // K = 1: [n][1] is [n]
    if (rc == POLEE_OK && zs_out) rc = vi->d_mu.download(ctx, zs_out, nm1);
    polee_vi_destroy(vi);
    return rc;
}

polee_status polee_sampler_draw(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                const float *z0, int32_t ndraws, uint64_t seed, float *xs)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!xs) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    DevBuf<float> d_all;
    POLEE_TRY(sampler_draw_device(t, mu, sigma, alpha, z0, ndraws, seed, d_all));
    return d_all.download(ctx, xs, (size_t)ndraws * t->n);
}

polee_status polee_sampler_initial_values(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                          const float *efflens, const float *z0, int32_t ndraws, uint64_t seed, float *x0)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!x0 || !efflens) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    DevBuf<float> d_all, d_l, d_x0;
    DevBuf<double> d_sums;
    POLEE_TRY(sampler_draw_device(t, mu, sigma, alpha, z0, ndraws, seed, d_all, 1e-10 /* LIKAP_Y_EPS */));
    const int64_t n = t->n;
    POLEE_TRY(d_l.upload(ctx, efflens, (size_t)n));
    POLEE_TRY(d_x0.alloc(ctx, (size_t)n));
    POLEE_TRY(d_sums.alloc(ctx, (size_t)ndraws));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d_sums.p, 0, sizeof(double) * ndraws, ctx->stream));
    hipLaunchKernelGGL(x0_sums_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 256), (unsigned)ndraws), dim3(256), 0,
                       ctx->stream, d_all.p, d_l.p, n, d_sums.p);
    hipLaunchKernelGGL(x0_mean_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_all.p, d_l.p, d_sums.p,
                       ndraws, n, d_x0.p);
    POLEE_KERNEL_CHECK(ctx);
    return d_x0.download(ctx, x0, (size_t)n);
}

polee_status polee_sampler_posterior_mean(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                          const float *z0, int32_t ndraws, uint64_t seed, float *pm)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!pm) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    DevBuf<float> d_all, d_pm;
    POLEE_TRY(sampler_draw_device(t, mu, sigma, alpha, z0, ndraws, seed, d_all));
    POLEE_TRY(d_pm.alloc(ctx, (size_t)t->n));
    hipLaunchKernelGGL(sampler_mean_kernel, dim3((unsigned)ceil_div(t->n, 256)), dim3(256), 0, ctx->stream, d_all.p,
                       ndraws, (int64_t)t->n, d_pm.p);
    POLEE_KERNEL_CHECK(ctx);
    return d_pm.download(ctx, pm, (size_t)t->n);
}

polee_status polee_sampler_quantiles(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                     const float *z0, int32_t ndraws, uint64_t seed, const double *qs, int32_t nq,
                                     float *quantiles)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!qs || !quantiles || nq < 1 || nq > SAMPLER_MAX_Q) return fail(ctx, POLEE_ERR_BAD_ARG, "1..%d quantiles", SAMPLER_MAX_Q);
    QuantileSpec spec{};
    spec.nq = nq;
    for (int i = 0; i < nq; ++i) {
        if (!(qs[i] >= 0.0 && qs[i] <= 1.0)) return fail(ctx, POLEE_ERR_BAD_ARG, "quantile %g outside [0, 1]", qs[i]);
        // Statistics.quantile, default (type 7): aleph = (N-1) q + 1, j = clamp(trunc(aleph), 1, N-1), gamma = aleph - j
        const double aleph = (double)(ndraws - 1) * qs[i] + 1.0;
        const int j = ndraws == 1 ? 1 : std::min(std::max((int)aleph, 1), ndraws - 1);
        spec.j[i] = j;
        spec.gamma[i] = std::min(std::max(aleph - j, 0.0), 1.0);
    }
    DevBuf<float> d_all, d_q;
    POLEE_TRY(sampler_draw_device(t, mu, sigma, alpha, z0, ndraws, seed, d_all));
    const int64_t n = t->n;
    POLEE_TRY(d_q.alloc(ctx, (size_t)nq * n));
    const bool in_lds = (size_t)ndraws * 64 * sizeof(float) <= 48 * 1024;
    hipLaunchKernelGGL(sampler_quantile_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64),
                       in_lds ? (size_t)ndraws * 64 * sizeof(float) : 0, ctx->stream, d_all.p, ndraws, n, spec,
                       in_lds ? 1 : 0, d_q.p);
    POLEE_KERNEL_CHECK(ctx);
    return d_q.download(ctx, quantiles, (size_t)nq * n);
}

}  // extern "C"
