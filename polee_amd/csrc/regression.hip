// The regression model's variational step, device resident.
// Replaces RNASeqLinearRegression.model_fn / variational_model_fn / fit (models/polee_regression.py:18-340):
// TensorFlow-Probability's JointDistributionCoroutine pair + tfp.vi.fit_surrogate_posterior(sample_size = 1,
// Adam(2e-3)) become hand-derived gradients of
//     loss = log q(z) - log p(z),   z one reparameterised draw of the surrogate posterior,
// with the approximate likelihood term supplied by polee_approx_logprob_device (approx.hip).
//
// One step:
//   noise    : Philox normals for every latent (or host-provided noise)           -> eps   [num_noise]
//   lse      : t_s = logsumexp_j qx_loc[s][j] (the scale-drift penalty's value)   -> lse   [S]
//   sample x : x = qx_loc + softplus(qx_softplus_scale) eps                       -> x     [S][n]
//   lik      : approximate likelihood of x with d/dx                              -> lp [S], glik [S][n]
//   data     : one thread per feature j over this rank's samples: observation-model sums (F+2 per column),
//              gradients of qx_loc / qx_softplus_scale, the samples' loss terms           -> stats [(F+2) n + 32]
//   (samples sharded over ranks: ONE all-reduce of stats, polee_regression_set_comm)
//   columns  : one thread per feature j: draws the horseshoe+ scales, w, x_bias, x_scale of its column,
//              evaluates its share of log q - log p and all per-column gradients from stats; block-reduces
//              the pieces shared by columns (global scale, distortion, mean-variance coefficients)
//   finish   : the few global parameters
//   adam     : Keras Adam over the flat parameter vector
// Everything is O((F + S) n) bytes per step (cache / latency bound): no roofline claim.
#include "common.hpp"
#include "comm_internal.hpp"
#include "rng.hpp"
#include "wave.hpp"

#include <cmath>

namespace polee {

constexpr int REG_MAXF = 16;    // factors (design-matrix columns)
constexpr int REG_MAXDEG = 32;  // kernel-regression hinges
constexpr int REG_BLOCK = 128;
constexpr int REG_SLOTS = 32;  // copies of every grid-wide accumulator (block b adds into copy b % 32): same-address
                               // float atomics from ~1.5 k blocks serialise, 32-way spreading removes that
constexpr float HALF_LOG2PI = 0.91893853320467274178f;

// Layout of the flat parameter / gradient vector and of the noise vector (include/polee_hip.h documents the order).
struct RegView {
    int32_t S, F, n, deg;
    int32_t use_distortion, point;
    float bias_loc0, bias_scale0, penalty;
    // > 0: x_scale ~ InverseGamma(fixed_ab, fixed_ab) instead of the kernel-regressed concentration / scale, and the
    // observation model stands alone (no sample scales, no scale-drift penalty, no likelihood term of its own): the
    // isoform block of the gene-isoform model (models/polee_regression.py:727-733), run with deg = 0
    float fixed_ab = 0.0f;
    // --- the joint model's variants (RNASeqJointLinearRegression, models/polee_regression.py:879-1283) ---
    int32_t levels = 2;       // 1: a horseshoe prior (one local scale level, :1009-1024) instead of horseshoe+; the local2 arrays stay in
                              //    the vector, unused (gradient 0)
    int32_t no_xs = 0;        // 1: no x_scale in this block (the splice-feature block: its observation lives on the transcripts)
    int32_t w_from_bias = 0;  // 1: the kernel-regression weights are functions of the SAMPLED bias (:1034-1035), not of x_bias_init
    float hc_scale = 1.0f;    // scale of the HalfCauchy prior on the mean-variance coefficients (10 in the joint model, :1037-1041)
    float bandwidth = 1.0f;
    const float *hinges = nullptr;  // (w_from_bias) device pointer, deg values
    __host__ __device__ int64_t Fn() const { return (int64_t)F * n; }
    __host__ __device__ int64_t o_dist() const { return 4; }
    __host__ __device__ int64_t o_conc() const { return 4 + (int64_t)F * deg; }
    __host__ __device__ int64_t o_scc() const { return o_conc() + deg; }
    __host__ __device__ int64_t o_cols() const { return o_scc() + deg; }  // 10 arrays [F][n]
    __host__ __device__ int64_t o_bias_loc() const { return o_cols() + 10 * Fn(); }
    __host__ __device__ int64_t o_bias_s() const { return o_bias_loc() + n; }
    __host__ __device__ int64_t o_xs_loc() const { return o_bias_loc() + 2 * (int64_t)n; }
    __host__ __device__ int64_t o_xs_s() const { return o_bias_loc() + 3 * (int64_t)n; }
    __host__ __device__ int64_t o_qx_loc() const { return o_bias_loc() + 4 * (int64_t)n; }
    __host__ __device__ int64_t o_qx_s() const { return o_qx_loc() + (int64_t)S * n; }
    __host__ __device__ int64_t num_params() const { return o_qx_s() + (int64_t)S * n; }
    // noise: 2 global, 5 arrays [F][n], x_bias [n], x_scale [n], x [S][n]
    __host__ __device__ int64_t e_cols() const { return 2; }
    __host__ __device__ int64_t e_bias() const { return 2 + 5 * Fn(); }
    __host__ __device__ int64_t e_xs() const { return e_bias() + n; }
    __host__ __device__ int64_t e_x() const { return e_bias() + 2 * (int64_t)n; }
    __host__ __device__ int64_t num_noise() const { return e_x() + (int64_t)S * n; }
    __host__ __device__ int num_red() const { return 1 + F * deg + 2 * deg; }
};

// The column kernels are VALU-bound (rocprofv3: ~7.4 k VALU instructions per thread with libm's exp / log / log1p and
// IEEE division), so the elementwise math uses the hardware transcendentals (v_exp_f32, v_log_f32, v_rcp_f32,
// v_sqrt_f32: 1 ulp each) -- errors of ~1e-6 relative, far inside the 1e-4 parity tolerance.
__device__ inline float fexp(float x) { return __expf(x); }
__device__ inline float flog(float x) { return __logf(x); }
__device__ inline float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// log1p(t), 0 <= t <= 1, with full relative accuracy for small t (log(1 + t) would round 1 + t)
__device__ inline float flog1p01(float t) { return t < 1e-3f ? t * (1.0f - t * (0.5f - t * (1.0f / 3.0f))) : flog(1.0f + t); }
__device__ inline float softplusf(float x) { return fmaxf(x, 0.0f) + flog1p01(fexp(-fabsf(x))); }
__device__ inline float sigmoidf(float x) { return frcp(1.0f + fexp(-x)); }
// lgamma(x) and psi(x), x > 0, together: the recurrences lgamma(x) = lgamma(x+1) - log x, psi(x) = psi(x+1) - 1/x up to
// x >= 8 (one log of the running product), then the Stirling / asymptotic series (truncation < 1e-8 at x = 8)
__device__ inline void lgamma_digamma(float x, float &lg, float &psi)
{
    float prod = 1.0f, r = 0.0f;
    while (x < 8.0f) {
        prod *= x;
        r -= frcp(x);
        x += 1.0f;
    }
    const float i = frcp(x), i2 = i * i, lx = flog(x);
    psi = r + lx - 0.5f * i - i2 * (1.0f / 12.0f - i2 * (1.0f / 120.0f - i2 * (1.0f / 252.0f)));
    lg = (x - 0.5f) * lx - x + HALF_LOG2PI + i * (1.0f / 12.0f - i2 * (1.0f / 360.0f - i2 * (1.0f / 1260.0f))) - flog(prod);
}

// one draw of SoftplusNormal(loc, softplus(sraw)) (src/polee.py:24-33) and its share of log q
struct SpDraw {
    float z, sg, eps, s, sgs, logq;
};
__device__ inline SpDraw sp_draw(float loc, float sraw, float eps)
{
    SpDraw d;
    d.eps = eps;
    d.s = softplusf(sraw);
    d.sgs = sigmoidf(sraw);
    const float u = loc + d.s * eps;
    d.z = softplusf(u);
    d.sg = sigmoidf(u);
    d.logq = -0.5f * eps * eps - flog(d.s) - HALF_LOG2PI + softplusf(-u);  // - log sigmoid(u)
    return d;
}
// G = d(-log p)/dz  ->  d loss / d loc, d loss / d sraw
__device__ inline void sp_grad(const SpDraw &d, float G, float &gloc, float &gs)
{
    const float a = G * d.sg - (1.0f - d.sg);
    gloc = a;
    gs = (a * d.eps - frcp(d.s)) * d.sgs;
}
// -log InverseGamma(0.5, 0.5)(z), -log HalfNormal(1)(z)
__device__ inline float nlp_ig_half(float z)
{
    return -(0.5f * -0.69314718055994530942f - 0.57236494292470008707f - 1.5f * flog(z) - 0.5f * frcp(z));
}
__device__ inline float nlp_halfnormal(float z) { return 0.22579135264472743236f + 0.5f * z * z; }  // -0.5 log(2/pi)

__device__ inline float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Device-side clock of the fit, so that a whole step is one replayable hipGraph: tick[0] = Adam step t, tick[1] =
// slot of the loss trace; seed_dev[0] = seed of the fit, lr_t[0] = Adam's step size at t.
__global__ void reg_tick_kernel(uint32_t *tick, float *lr_t, float lr)
{
    const double t = (double)(++tick[0]);
    lr_t[0] = (float)((double)lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
}

// seed / step: immediates, or (tick != nullptr) read from the device clock
__global__ void reg_noise_kernel(int64_t count, uint64_t seed, uint32_t step, const uint64_t *seed_dev,
                                 const uint32_t *tick, uint64_t salt, float *eps, uint32_t tick_ahead = 0u)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= count) return;
    if (tick) {
        seed = seed_dev[0];
        step = tick[0] + tick_ahead;  // (tick_ahead = 1: the clock is advanced later in the step, reg_finish_kernel)
    }
    seed ^= salt;
    float z[4];
    philox_randn4(seed ^ 0x7265677265737369ull, step, (uint32_t)(q >> 32), (uint32_t)q, z);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (q * 4 + i < count) eps[q * 4 + i] = z[i];
}

// One launch per step for both noise segments and the draw of x (round 6: the tick, two noise launches and reg_sample_x_kernel
// were four ~5 us launches of a ~290 us step).  The step number is the device clock's NEXT value, tick[0] + 1: the clock
// itself is advanced by reg_finish_kernel, the step's last single-workgroup launch in front of Adam.  Segment 0 = the latents
// every rank shares (salt 0), segment 1 = the x noise of this rank's samples (salted): the same Philox blocks, in the same
// places, as two reg_noise_kernel launches.  A thread of segment 1 also writes x = qx_loc + softplus(qx_scale) eps for its
// four entries (x == nullptr: point estimates, no draw).
__global__ void reg_draw_kernel(RegView v, int64_t shared, int64_t own, const uint64_t *seed_dev, const uint32_t *tick, uint64_t salt,
                                const float *__restrict__ p, float *__restrict__ eps, float *__restrict__ x)
{
    const int64_t nq0 = (shared + 3) / 4;
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool seg1 = q >= nq0;
    if (seg1) q -= nq0;
    const int64_t count = seg1 ? own : shared;
    if (q * 4 >= count) return;
    const uint64_t seed = seed_dev[0] ^ (seg1 ? salt : (uint64_t)0);
    const uint32_t step = tick[0] + 1u;
    float z[4];
    philox_randn4(seed ^ 0x7265677265737369ull, step, (uint32_t)(q >> 32), (uint32_t)q, z);
    float *e = eps + (seg1 ? shared : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (q * 4 + i < count) {
            e[q * 4 + i] = z[i];
            if (seg1 && x) x[q * 4 + i] = p[v.o_qx_loc() + q * 4 + i] + softplusf(p[v.o_qx_s() + q * 4 + i]) * z[i];
        }
}

// t_s = logsumexp_j qx_loc[s][j]   (qx_sample_scale, models/polee_regression.py:300-301)
__global__ __launch_bounds__(1024) void reg_lse_kernel(RegView v, const float *p, float *lse)
{
    __shared__ float red[16];
    const float *row = p + v.o_qx_loc() + (int64_t)blockIdx.x * v.n;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < v.n; j += 1024) m = fmaxf(m, row[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    __syncthreads();
    float s = 0.0f;
    for (int j = threadIdx.x; j < v.n; j += 1024) s += expf(row[j] - m);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int i = 0; i < 16; ++i) t += red[i];
        lse[blockIdx.x] = m + logf(t);
    }
}

// The same once a nearby shift is known (the previous step's value: Adam moves qx_loc by ~lr per step), so that the
// row can be spread over many blocks: acc[s] += sum exp(qx_loc - lse_prev[s]); then lse = lse_prev + log acc.
__global__ __launch_bounds__(256) void reg_lse_accum_kernel(RegView v, const float *p, const float *lse, float *acc)
{
    __shared__ float red[4];
    const int s = blockIdx.y;
    const float *row = p + v.o_qx_loc() + (int64_t)s * v.n;
    const float c = lse[s];
    float t = 0.0f;
    const int64_t j0 = (int64_t)blockIdx.x * 4096, j1 = j0 + 4096 < v.n ? j0 + 4096 : v.n;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) t += expf(row[j] - c);
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&acc[s], red[0] + red[1] + red[2] + red[3]);
}
__global__ void reg_lse_finish_kernel(int S, float *lse, float *acc)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    lse[s] += logf(acc[s]);
    acc[s] = 0.0f;
}

__global__ void reg_sample_x_kernel(RegView v, const float *p, const float *eps, float *x)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)v.S * v.n) return;
    x[i] = p[v.o_qx_loc() + i] + softplusf(p[v.o_qx_s() + i]) * eps[v.e_x() + i];
}

// Likelihood of RNASeqNormalTranscriptLinearRegression (models/polee_regression.py:490-531): the point estimates
// v [S][n] ~ Normal(log softmax(x), sigma).  One block per sample: lp[s] and glik = d lp / d x.
__global__ __launch_bounds__(1024) void reg_normal_lik_kernel(int n, const float *__restrict__ x,
                                                               const float *__restrict__ v,
                                                               const float *__restrict__ sigma, float *lp, float *glik)
{
    __shared__ float red[16];
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *xr = x + (int64_t)s * n, *vr = v + (int64_t)s * n, *sr = sigma + (int64_t)s * n;
    float *gr = glik + (int64_t)s * n;
    auto block_sum = [&](float val) {
        val = wave_sum(val);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = val;
        __syncthreads();
        float t = 0.0f;
        for (int i = 0; i < 16; ++i) t += red[i];
        return t;
    };
    float m = -INFINITY;
    for (int j = tid; j < n; j += 1024) m = fmaxf(m, xr[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    float se = 0.0f;
    for (int j = tid; j < n; j += 1024) se += expf(xr[j] - m);
    const float lse = m + logf(block_sum(se));
    float sr_sum = 0.0f, l = 0.0f;
    for (int j = tid; j < n; j += 1024) {
        const float is = 1.0f / sr[j], d = (vr[j] - (xr[j] - lse)) * is;
        sr_sum += d * is;
        l += -0.5f * d * d - logf(sr[j]) - HALF_LOG2PI;
    }
    const float R = block_sum(sr_sum);
    const float L = block_sum(l);
    if (tid == 0) lp[s] = L;
    for (int j = tid; j < n; j += 1024) {
        const float is = 1.0f / sr[j];
        gr[j] = (vr[j] - (xr[j] - lse)) * is * is - expf(xr[j] - lse) * R;
    }
}

// ---- gene-level model (RNASeqGeneLinearRegression, models/polee_regression.py:533-600): the features are genes and
// the likelihood is reached through within-gene isoform log-expression x_isoform [S][nt] with
//   x_isoform_mean ~ Normal(0, 2) [nt],  x_isoform ~ Normal(x_isoform_mean, 1),  both with Normal surrogates.
// Isoform block of parameters: mean_loc [nt], mean_softplus_scale [nt], iso_loc [S][nt], iso_softplus_scale [S][nt];
// noise: mean [nt], iso [S][nt].
__global__ void reg_iso_sample_kernel(int S, int nt, const float *ip, const float *ieps, float *xi)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, snt = (int64_t)S * nt;
    if (i >= snt) return;
    xi[i] = ip[2 * (int64_t)nt + i] + softplusf(ip[2 * (int64_t)nt + snt + i]) * ieps[nt + i];
}
// gi = d lp / d x_isoform (from the gene-level likelihood); thread per transcript
__global__ __launch_bounds__(256) void reg_iso_grad_kernel(int S, int nt, const float *__restrict__ ip,
                                                           const float *__restrict__ ieps,
                                                           const float *__restrict__ gi, float *__restrict__ ig,
                                                           float *loss_slots)
{
    const int64_t ii = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, snt = (int64_t)S * nt;
    const bool live = ii < nt;
    const int64_t i = live ? ii : nt - 1;
    const float *loc = ip + 2 * (int64_t)nt, *sr = loc + snt;
    const float ms_raw = ip[nt + i], ms = softplusf(ms_raw), me = ieps[i];
    const float m = ip[i] + ms * me;
    float loss = -0.5f * me * me - flog(ms) - HALF_LOG2PI;               // log q(mean)
    loss += 0.125f * m * m + 0.69314718055994530942f + HALF_LOG2PI;      // -log Normal(0, 2)(mean)
    float Gm = 0.25f * m;
    for (int s = 0; s < S; ++s) {
        const int64_t o = (int64_t)s * nt + i;
        const float sraw = sr[o], sx = softplusf(sraw), e = ieps[nt + o];
        const float d = loc[o] + sx * e - m;
        loss += 0.5f * d * d + HALF_LOG2PI;                              // -log Normal(mean, 1)(x_isoform)
        loss += -0.5f * e * e - flog(sx) - HALF_LOG2PI;                  // log q(x_isoform)
        Gm -= d;
        const float Gx = d - gi[o];
        if (live) {
            ig[2 * (int64_t)nt + o] = Gx;
            ig[2 * (int64_t)nt + snt + o] = (Gx * e - frcp(sx)) * sigmoidf(sraw);
        }
    }
    if (live) {
        ig[i] = Gm;
        ig[nt + i] = (Gm * me - frcp(ms)) * sigmoidf(ms_raw);
    }
    loss = wave_sum_to_lane63(live ? loss : 0.0f);
    if ((threadIdx.x & 63) == 63) atomicAdd(&loss_slots[blockIdx.x % REG_SLOTS], loss);
}

// ---- joint model (RNASeqJointLinearRegression, models/polee_regression.py:879-1283): beside the gene block a SPLICE block --
// horseshoe coefficients w_splice [F][P] and a bias ~ Normal(0, 10) over P splice features (:1062-1087), whose linear
// predictor reaches the transcripts through the 0/1 feature matrix (:1089-1108): x_iso_loc[s][t] = sum of (F w + b)[s][p] over
// the features p of transcript t; x_iso_scale ~ HalfCauchy(0, 1) [nt] (:1110-1112), x_iso ~ Normal(x_iso_loc, x_iso_scale)
// (:1114-1116), and the gene-level likelihood of (x_gene, x_iso) (:1118-1121).  The splice features' coefficients are a
// RegView over P "columns" (levels = 1, no_xs) run through the column / finish kernels; the transcripts' part -- x_iso_scale
// (SoftplusNormal surrogate) and x_iso (Normal surrogate) -- is the kernel below.  Transcript part of the parameter block:
// x_iso_scale_loc [nt], x_iso_scale_softplus_scale [nt], x_iso_loc [S][nt], x_iso_softplus_scale [S][nt]; noise: scale [nt],
// x_iso [S][nt].
__global__ void reg_joint_sample_kernel(int S, int nt, const float *jp, const float *jeps, float *xi)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, snt = (int64_t)S * nt;
    if (i >= snt) return;
    xi[i] = jp[2 * (int64_t)nt + i] + softplusf(jp[2 * (int64_t)nt + snt + i]) * jeps[nt + i];
}
// mu[s][p] = x_splice_bias[p] + sum_f F[s][f] w_splice[f][p] at the block's draw (thread per feature)
__global__ void reg_joint_mean_kernel(RegView vs, int S, const float *__restrict__ sp, const float *__restrict__ seps,
                                      const float *__restrict__ design, float *__restrict__ mu)
{
    const int64_t pidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= vs.n) return;
    const int64_t Fn = vs.Fn();
    const float b = sp[vs.o_bias_loc() + pidx] + softplusf(sp[vs.o_bias_s() + pidx]) * seps[vs.e_bias() + pidx];
    float w[REG_MAXF];
    for (int f = 0; f < vs.F; ++f) {
        const int64_t idx = (int64_t)f * vs.n + pidx;
        w[f] = sp[vs.o_cols() + 8 * Fn + idx] + softplusf(sp[vs.o_cols() + 9 * Fn + idx]) * seps[vs.e_cols() + 4 * Fn + idx];
    }
    for (int s = 0; s < S; ++s) {
        float m = b;
        for (int f = 0; f < vs.F; ++f) m += design[s * vs.F + f] * w[f];
        mu[(int64_t)s * vs.n + pidx] = m;
    }
}
// thread per transcript: x_iso_scale's draw, the observation term of its S values, log q of both, their gradients (gi = d lp /
// d x_iso from the gene-level likelihood); resid[s][t] = (x_iso - x_iso_loc) / x_iso_scale^2 goes on to the features
__global__ __launch_bounds__(256) void reg_joint_iso_kernel(int S, int nt, int P, const float *__restrict__ jp,
                                                            const float *__restrict__ jeps, const float *__restrict__ gi,
                                                            const float *__restrict__ mu, const int32_t *__restrict__ t_ptr,
                                                            const int32_t *__restrict__ t_feat, float *__restrict__ jg,
                                                            float *__restrict__ resid, float *loss_slots)
{
    const int64_t ii = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, snt = (int64_t)S * nt;
    const bool live = ii < nt;
    const int64_t i = live ? ii : nt - 1;
    const float *loc = jp + 2 * (int64_t)nt, *sr = loc + snt;
    const SpDraw xs = sp_draw(jp[i], jp[nt + i], jeps[i]);
    const float z = xs.z, iz = frcp(z), iz2 = iz * iz, lz = flog(z);
    float loss = xs.logq + 0.45158270528945486473f + log1pf(z * z);  // log q - log HalfCauchy(0, 1)
    float Gz = 2.0f * z / (1.0f + z * z);
    const int32_t f0 = t_ptr[i], f1 = t_ptr[i + 1];
    for (int s = 0; s < S; ++s) {
        const int64_t o = (int64_t)s * nt + i;
        float m = 0.0f;
        for (int32_t q = f0; q < f1; ++q) m += mu[(int64_t)s * P + t_feat[q]];
        const float sraw = sr[o], sx = softplusf(sraw), e = jeps[nt + o];
        const float d = loc[o] + sx * e - m;
        const float a = d * iz2;
        loss += 0.5f * d * a + lz + HALF_LOG2PI;              // -log Normal(x_iso_loc, x_iso_scale)(x_iso)
        loss += -0.5f * e * e - flog(sx) - HALF_LOG2PI;       // log q(x_iso)
        Gz += iz - d * a * iz;
        const float Gx = a - gi[o];
        if (live) {
            jg[2 * (int64_t)nt + o] = Gx;
            jg[2 * (int64_t)nt + snt + o] = (Gx * e - frcp(sx)) * sigmoidf(sraw);
            resid[o] = a;
        }
    }
    if (live) {
        float a, c;
        sp_grad(xs, Gz, a, c);
        jg[i] = a;
        jg[nt + i] = c;
    }
    loss = wave_sum_to_lane63(live ? loss : 0.0f);
    if ((threadIdx.x & 63) == 63) atomicAdd(&loss_slots[blockIdx.x % REG_SLOTS], loss);
}
// thread per splice feature: what the transcripts say about its column (the `stats` rows the column kernel reads)
__global__ void reg_joint_agg_kernel(int S, int nt, int P, int F, const float *__restrict__ resid,
                                     const float *__restrict__ design, const int32_t *__restrict__ p_ptr,
                                     const int32_t *__restrict__ p_trans, float *__restrict__ stats)
{
    const int64_t pidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= P) return;
    float gacc[REG_MAXF], sum_a = 0.0f;
    for (int f = 0; f < F; ++f) gacc[f] = 0.0f;
    const int32_t q0 = p_ptr[pidx], q1 = p_ptr[pidx + 1];
    for (int s = 0; s < S; ++s) {
        float a = 0.0f;
        for (int32_t q = q0; q < q1; ++q) a += resid[(int64_t)s * nt + p_trans[q]];
        sum_a += a;
        for (int f = 0; f < F; ++f) gacc[f] -= design[s * F + f] * a;
    }
    for (int f = 0; f < F; ++f) stats[(int64_t)f * P + pidx] = gacc[f];
    stats[(int64_t)F * P + pidx] = sum_a;
    stats[(int64_t)(F + 1) * P + pidx] = 0.0f;
}

// ---- data pass: what the S (local) samples say about each column ------------------------------------------
// stats [F+2][n] + REG_SLOTS:  rows 0..F-1  sum_s design[s][f] d(-log p_x)/d x_loc[s][j];  row F  sum_s (x - mu)/x_scale^2;
// row F+1  sum_s d(-log p_x)/d x_scale;  last REG_SLOTS values (summed by the reader)  loss terms of the samples (observation model, log q of x,
// scale-drift penalty, -likelihood).  These are sums over samples: with the samples sharded over ranks ONE all-reduce
// of this buffer is the only exchange of a step.  Also writes d loss / d qx_loc, qx_softplus_scale.
// (FT, DT: compile-time F and degree, 0 = run-time; fixed trip counts let the compiler issue a column's loads together)
template <int FT, int DT>
__global__ __launch_bounds__(REG_BLOCK) void reg_data_kernel(RegView v, const float *__restrict__ p,
                                                             const float *__restrict__ eps,
                                                             const float *__restrict__ design,
                                                             const float *__restrict__ W, const float *__restrict__ ss,
                                                             const float *__restrict__ x, const float *__restrict__ glik,
                                                             const float *__restrict__ lse, const float *__restrict__ lp,
                                                             float *__restrict__ g, float *__restrict__ stats)
{
    __shared__ float s_weff[FT ? FT : REG_MAXF][REG_BLOCK], s_gacc[FT ? FT : REG_MAXF][REG_BLOCK];
    const int tid = threadIdx.x;
    const int64_t jj = (int64_t)blockIdx.x * REG_BLOCK + tid;
    const bool live = jj < v.n;
    const int64_t j = live ? jj : v.n - 1;  // dead lanes recompute the last column and contribute nothing
    constexpr int UF = FT ? FT : 1, UD = DT ? DT : 1;  // unroll counts
    const int F = FT ? FT : v.F, deg = DT ? DT : v.deg, n = v.n;
    const int64_t Fn = v.Fn();
#pragma unroll UF
    for (int f = 0; f < F; ++f) {
        const int64_t idx = (int64_t)f * n + j;
        const float w = p[v.o_cols() + 8 * Fn + idx] + softplusf(p[v.o_cols() + 9 * Fn + idx]) * eps[v.e_cols() + 4 * Fn + idx];
        float wd = 0.0f;
        if (v.use_distortion) {
#pragma unroll UD
            for (int d = 0; d < deg; ++d) wd += p[v.o_dist() + f * deg + d] * W[(int64_t)d * n + j];
        }
        s_weff[f][tid] = w + wd;
        s_gacc[f][tid] = 0.0f;
    }
    const float b = p[v.o_bias_loc() + j] + softplusf(p[v.o_bias_s() + j]) * eps[v.e_bias() + j];
    const float xsz = softplusf(p[v.o_xs_loc() + j] + softplusf(p[v.o_xs_s() + j]) * eps[v.e_xs() + j]);
    const float inv = frcp(xsz), inv2 = inv * inv, lxs = flog(xsz);
    const float ipen2 = 1.0f / (v.penalty * v.penalty);
    const bool sub = v.fixed_ab > 0.0f;  // (then ss, x, lse are null: x is recomputed from its surrogate)
    float loss = 0.0f, sum_a = 0.0f, sum_xs = 0.0f;
#pragma unroll 2  // (the samples' loads are independent: two in flight per lane)
    for (int s = 0; s < v.S; ++s) {
        float xl = b;
#pragma unroll UF
        for (int f = 0; f < F; ++f) xl += design[s * F + f] * s_weff[f][tid];
        const int64_t sj = (int64_t)s * n + j;
        const float xv = v.point ? p[v.o_qx_loc() + sj]
                                 : (sub ? p[v.o_qx_loc() + sj] + softplusf(p[v.o_qx_s() + sj]) * eps[v.e_x() + sj] : x[sj]);
        const float d = xv - (xl - (sub ? 0.0f : ss[s]));
        const float a = d * inv2;
        loss += 0.5f * d * a + lxs + HALF_LOG2PI;
        sum_xs += inv - d * a * inv;
        sum_a += a;
#pragma unroll UF
        for (int f = 0; f < F; ++f) s_gacc[f][tid] -= design[s * F + f] * a;
        float gl = 0.0f, gs = 0.0f;
        if (!v.point) {
            const float sraw = p[v.o_qx_s() + sj], sx = softplusf(sraw), e = eps[v.e_x() + sj];
            const float Gx = a - glik[sj];
            gl = sub ? Gx : Gx + lse[s] * ipen2 * fexp(p[v.o_qx_loc() + sj] - lse[s]);
            gs = (Gx * e - frcp(sx)) * sigmoidf(sraw);
            loss += -0.5f * e * e - flog(sx) - HALF_LOG2PI;
        }
        if (live) g[v.o_qx_loc() + sj] = gl, g[v.o_qx_s() + sj] = gs;
    }
    if (live) {
#pragma unroll UF
        for (int f = 0; f < F; ++f) stats[(int64_t)f * n + j] = s_gacc[f][tid];
        stats[(int64_t)F * n + j] = sum_a;
        stats[(int64_t)(F + 1) * n + j] = sum_xs;
    }
    loss = wave_sum(live ? loss : 0.0f);
    if (blockIdx.x == 0 && tid == 0 && !v.point && !sub)
        for (int s = 0; s < v.S; ++s) {  // per-sample terms: scale-drift penalty, approximate likelihood
            const float t = lse[s] / v.penalty;
            loss += 0.5f * t * t + logf(v.penalty) + HALF_LOG2PI - (lp ? lp[s] : 0.0f);
        }
    if ((tid & 63) == 0) atomicAdd(&stats[(int64_t)(F + 2) * n + (blockIdx.x % REG_SLOTS)], loss);
}

// ---- prior pass: everything that does not depend on which samples a rank holds, combined with the (summed) stats
// acc: loss (double [REG_SLOTS]); small [REG_SLOTS][num_red]: [0] sum (1 - r^2), then d/d(distortion_c) [F][deg], then the sums feeding
// d/d(concentration_c) [deg] and d/d(scale_c) [deg]
template <int FT, int DT>
__global__ __launch_bounds__(REG_BLOCK) void reg_cols_kernel(RegView v, const float *__restrict__ p,
                                                             const float *__restrict__ eps,
                                                             const float *__restrict__ W,
                                                             const float *__restrict__ stats, float *__restrict__ g,
                                                             double *loss_acc, float *small)
{
    __shared__ float s_red[1 + REG_MAXF * REG_MAXDEG + 2 * REG_MAXDEG];
    __shared__ float s_cc[REG_MAXDEG], s_sc[REG_MAXDEG];
    const int tid = threadIdx.x;
    const int64_t jj = (int64_t)blockIdx.x * REG_BLOCK + tid;
    const bool live = jj < v.n;
    const int64_t j = live ? jj : v.n - 1;
    constexpr int UF = FT ? FT : 1, UD = DT ? DT : 1;  // unroll counts
    const int F = FT ? FT : v.F, deg = DT ? DT : v.deg, n = v.n;
    const int64_t Fn = v.Fn();
    const int nred = v.num_red();
    const float lv = live ? 1.0f : 0.0f;
    const int lane = tid & 63;
    for (int i = tid; i < nred; i += REG_BLOCK) s_red[i] = 0.0f;
    if (tid < deg) {
        s_cc[tid] = softplusf(p[v.o_conc() + tid]);
        s_sc[tid] = softplusf(p[v.o_scc() + tid]);
    }
    __syncthreads();

    float loss = 0.0f, S1 = 0.0f;
    // global horseshoe scale (every thread recomputes the two scalars)
    const SpDraw gv = sp_draw(p[0], p[1], eps[0]), gn = sp_draw(p[2], p[3], eps[1]);
    const float gscale = gn.z * fsqrt(gv.z);

    // ---- horseshoe+ scales and w of every factor
#pragma unroll UF
    for (int f = 0; f < F; ++f) {
        const int64_t idx = (int64_t)f * n + j;
        const float *pc = p + v.o_cols() + idx;
        const float *ec = eps + v.e_cols() + idx;
        float *gc = g + v.o_cols() + idx;
        const SpDraw l1v = sp_draw(pc[0 * Fn], pc[1 * Fn], ec[0 * Fn]);
        const SpDraw l1n = sp_draw(pc[2 * Fn], pc[3 * Fn], ec[1 * Fn]);
        SpDraw l2v = sp_draw(pc[4 * Fn], pc[5 * Fn], ec[2 * Fn]);
        SpDraw l2n = sp_draw(pc[6 * Fn], pc[7 * Fn], ec[3 * Fn]);
        const bool two = v.levels > 1;
        if (!two) {  // horseshoe: no second local scale
            l2v.z = l2n.z = 1.0f;
            l2v.logq = l2n.logq = 0.0f;
        }
        const float sraw_w = pc[9 * Fn], s_w = softplusf(sraw_w), e_w = ec[4 * Fn];
        const float w = pc[8 * Fn] + s_w * e_w;
        const float sw = (l1n.z * fsqrt(l1v.z)) * (l2n.z * fsqrt(l2v.z)) * gscale;
        const float isw = frcp(sw), r = w * isw, q = 1.0f - r * r;
        S1 += q;
        loss += l1v.logq + l1n.logq + l2v.logq + l2n.logq + (-0.5f * e_w * e_w - flog(s_w) - HALF_LOG2PI);
        loss += nlp_ig_half(l1v.z) + nlp_halfnormal(l1n.z) + (two ? nlp_ig_half(l2v.z) + nlp_halfnormal(l2n.z) : 0.0f);
        loss += 0.5f * r * r + flog(sw) + HALF_LOG2PI;
        float a, b;
        float iz = frcp(l1v.z);
        sp_grad(l1v, iz * (1.5f + 0.5f * q - 0.5f * iz), a, b);  // 1.5/z - 0.5/z^2 + 0.5 q/z
        if (live) gc[0 * Fn] = a, gc[1 * Fn] = b;
        sp_grad(l1n, l1n.z + q * frcp(l1n.z), a, b);
        if (live) gc[2 * Fn] = a, gc[3 * Fn] = b;
        iz = frcp(l2v.z);
        sp_grad(l2v, iz * (1.5f + 0.5f * q - 0.5f * iz), a, b);
        if (live) gc[4 * Fn] = two ? a : 0.0f, gc[5 * Fn] = two ? b : 0.0f;
        sp_grad(l2n, l2n.z + q * frcp(l2n.z), a, b);
        if (live) gc[6 * Fn] = two ? a : 0.0f, gc[7 * Fn] = two ? b : 0.0f;
        const float gacc = stats[idx];
        const float Gw = r * isw + gacc;
        if (live) {
            gc[8 * Fn] = Gw;
            gc[9 * Fn] = (Gw * e_w - frcp(s_w)) * sigmoidf(sraw_w);
        }
        if (v.use_distortion) {
#pragma unroll UD
            for (int d = 0; d < deg; ++d) {
                const float t = wave_sum_to_lane63(lv * W[(int64_t)d * n + j] * gacc);
                if (lane == 63) atomicAdd(&s_red[1 + f * deg + d], t);
            }
        }
    }

    // ---- x_bias, x_scale
    const float s_b = softplusf(p[v.o_bias_s() + j]), e_b = eps[v.e_bias() + j];
    const float b = p[v.o_bias_loc() + j] + s_b * e_b;
    loss += -0.5f * e_b * e_b - flog(s_b) - HALF_LOG2PI;
    const float db = (b - v.bias_loc0) / v.bias_scale0;
    float Gb = db / v.bias_scale0 - stats[(int64_t)F * n + j];
    loss += 0.5f * db * db + logf(v.bias_scale0) + HALF_LOG2PI;
    float g_alpha = 0.0f, g_beta = 0.0f;
    // the kernel-regression weights of this column: precomputed from x_bias_init (W), or -- the joint model -- functions of
    // the sampled bias b (src/polee.py:36-47): k_d = clip(exp(-((b - h_d) / bw)^2), 1e-10, 1), w_d = k_d / sum k
    // (only the run-time-shape instance <0, 0> carries this variant: the fixed-shape instances of the transcript model keep
    // their register budget; the host launches <0, 0> for a view with w_from_bias)
    constexpr bool WB = FT == 0 && DT == 0;
    float wcol[WB ? REG_MAXDEG : 1], dl[WB ? REG_MAXDEG : 1];
    const bool wfb = WB && v.w_from_bias;
    if (wfb) {
        float tot = 0.0f;
        for (int d = 0; d < deg; ++d) {
            const float u = (b - v.hinges[d]) / v.bandwidth;
            const float k0 = fexp(-u * u);
            const bool clipped = k0 < 1e-10f;  // (the upper clip at 1 is never active: exp(-u^2) <= 1)
            wcol[d] = clipped ? 1e-10f : k0;
            dl[d] = clipped ? 0.0f : -2.0f * u / v.bandwidth;  // d log k_d / d b
            tot += wcol[d];
        }
        const float it = frcp(tot);
        for (int d = 0; d < deg; ++d) wcol[d] *= it;
    }
    if (!v.no_xs) {
        const SpDraw xs = sp_draw(p[v.o_xs_loc() + j], p[v.o_xs_s() + j], eps[v.e_xs() + j]);
        loss += xs.logq;
        float alpha = 0.0f, beta = 0.0f;
        if (wfb) {
            for (int d = 0; d < deg; ++d) {
                alpha += s_cc[d] * wcol[d];
                beta += s_sc[d] * wcol[d];
            }
        } else {
#pragma unroll UD
            for (int d = 0; d < deg; ++d) {
                const float wdj = W[(int64_t)d * n + j];
                alpha += s_cc[d] * wdj;
                beta += s_sc[d] * wdj;
            }
        }
        if (v.fixed_ab > 0.0f) alpha = beta = v.fixed_ab;
        const float inv = frcp(xs.z), inv2 = inv * inv, lxs = flog(xs.z);
        const float Gxs = (alpha + 1.0f) * inv - beta * inv2 + stats[(int64_t)(F + 1) * n + j];
        const float lbeta = flog(beta);
        float lg_alpha, psi_alpha;
        lgamma_digamma(alpha, lg_alpha, psi_alpha);
        loss += -(alpha * lbeta - lg_alpha - (alpha + 1.0f) * lxs - beta * inv);
        g_alpha = -lbeta + psi_alpha + lxs;
        g_beta = -alpha * frcp(beta) + inv;
        if (wfb) {  // d alpha / d b = sum_d cc_d w_d (l_d - lbar), lbar = sum_e w_e l_e; the same for beta
            float lbar = 0.0f, da = 0.0f, dbt = 0.0f;
            for (int d = 0; d < deg; ++d) lbar += wcol[d] * dl[d];
            for (int d = 0; d < deg; ++d) {
                da += s_cc[d] * wcol[d] * (dl[d] - lbar);
                dbt += s_sc[d] * wcol[d] * (dl[d] - lbar);
            }
            Gb += g_alpha * da + g_beta * dbt;
        }
        if (live) {
            float a, c;
            sp_grad(xs, Gxs, a, c);
            g[v.o_xs_loc() + j] = a;
            g[v.o_xs_s() + j] = c;
        }
    } else if (live) {
        g[v.o_xs_loc() + j] = 0.0f;
        g[v.o_xs_s() + j] = 0.0f;
    }
    if (live) {
        g[v.o_bias_loc() + j] = Gb;
        g[v.o_bias_s() + j] = (Gb * e_b - frcp(s_b)) * sigmoidf(p[v.o_bias_s() + j]);
    }

    // ---- block sums of what the columns share
#pragma unroll UD
    for (int d = 0; d < deg; ++d) {
        const float wdj = lv * (wfb ? wcol[WB ? d : 0] : W[(int64_t)d * n + j]);
        const float ta = wave_sum_to_lane63(wdj * g_alpha), tb = wave_sum_to_lane63(wdj * g_beta);
        if (lane == 63) {
            atomicAdd(&s_red[1 + F * deg + d], ta);
            atomicAdd(&s_red[1 + F * deg + deg + d], tb);
        }
    }
    S1 = wave_sum_to_lane63(lv * S1);
    loss = wave_sum_to_lane63(lv * loss);
    const int slot = blockIdx.x % REG_SLOTS;
    if (lane == 63) {
        atomicAdd(&s_red[0], S1);
        atomicAdd(&loss_acc[slot], (double)loss);
    }
    __syncthreads();
    for (int i = tid; i < nred; i += REG_BLOCK) atomicAdd(&small[(int64_t)slot * nred + i], s_red[i]);
}

// the global horseshoe scale and the distortion / mean-variance coefficients (one wave; lane i takes coefficient i)
// (it leaves every accumulator it has read at zero for the next step: no memsets between steps)
__global__ __launch_bounds__(64) void reg_finish_kernel(RegView v, const float *p, const float *eps, float *small,
                                                        float *stats, double *loss_acc, float *g, float *loss_out,
                                                        const float *extra_loss, uint32_t *tick, float *lr_t, float lr)
{
    const int lane = threadIdx.x, nred = v.num_red();
    if (tick && lane == 0) {  // the device clock of fit(): this step's number and Adam's bias-corrected rate (reg_tick_kernel)
        const double t = (double)(++tick[0]);
        lr_t[0] = (float)((double)lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
    }
    auto slots = [&](int i) {  // sum of accumulator i over its copies
        float t = 0.0f;
        for (int k = 0; k < REG_SLOTS; ++k) {
            t += small[(int64_t)k * nred + i];
            small[(int64_t)k * nred + i] = 0.0f;
        }
        return t;
    };
    double loss = 0.0;
    if (lane < REG_SLOTS) {  // the columns' terms + the samples' terms (summed over ranks)
        loss = loss_acc[lane] + (double)stats[(int64_t)(v.F + 2) * v.n + lane];
        loss_acc[lane] = 0.0;
        stats[(int64_t)(v.F + 2) * v.n + lane] = 0.0f;
    }
    if (lane == 0) {
        const float S1 = slots(0);
        const SpDraw gv = sp_draw(p[0], p[1], eps[0]), gn = sp_draw(p[2], p[3], eps[1]);
        loss += gv.logq + gn.logq + nlp_ig_half(gv.z) + nlp_halfnormal(gn.z);
        sp_grad(gv, 1.5f / gv.z - 0.5f / (gv.z * gv.z) + 0.5f * S1 / gv.z, g[0], g[1]);
        sp_grad(gn, gn.z + S1 / gn.z, g[2], g[3]);
    }
    for (int i = lane; i < v.F * v.deg; i += 64) {
        const float c = p[v.o_dist() + i];
        if (v.use_distortion) {
            g[v.o_dist() + i] = 2.0f * c / (0.01f + c * c) + slots(1 + i);
            loss += logf(3.14159265358979323846f * 0.1f) + log1pf(100.0f * c * c);
        } else
            g[v.o_dist() + i] = 0.0f;
    }
    for (int i = lane; i < 2 * v.deg; i += 64) {  // concentration_c then scale_c (adjacent in the vector)
        const int64_t o = v.o_conc() + i;
        const float c = softplusf(p[o]), hs = v.hc_scale;  // HalfCauchy(0, hs): -log p = -log(2 / (pi hs)) + log1p((c / hs)^2)
        g[o] = (2.0f * c / (hs * hs + c * c) + slots(1 + v.F * v.deg + i)) * sigmoidf(p[o]);
        loss += 0.45158270528945486473f + logf(hs) + log1pf((c / hs) * (c / hs));  // (0.4515... = -log(2/pi))
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o, 64);
    if (lane == 0) loss_out[0] = (float)(loss + (extra_loss ? (double)extra_loss[0] : 0.0));
}

// tf.optimizers.Adam: theta -= lr sqrt(1 - b2^t) / (1 - b1^t) m / (sqrt(v) + eps)
__global__ void reg_adam_kernel(int64_t count, float *p, const float *g, float *m, float *vv, const float *lr_dev,
                                const float *loss, float *trace, uint32_t *tick)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && trace) trace[tick[1]++] = loss[0];  // (also when nothing is trainable: count may be 0)
    if (i >= count) return;
    const float lr_t = lr_dev[0];
    const float gi = g[i];
    const float mi = 0.9f * m[i] + 0.1f * gi;
    const float vi = 0.999f * vv[i] + 0.001f * gi * gi;
    m[i] = mi;
    vv[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + 1e-7f);
}

// ---- classify (models/polee_regression.py:342-413): the design matrix of the testing samples is a latent variable, so the loss needs
// its gradient with respect to F[s][f].  F enters through the observation model only, x[s][j] ~ Normal(sum_f F[s][f] w_eff[f][j] + bias[j]
// - scale[s], x_scale[j]):  d(-log p)/dF[s][f] = -sum_j a[s][j] w_eff[f][j]  with a = (x - mu) / x_scale^2 -- the quantities of
// reg_data_kernel, recomputed here for the same draw (a kernel of its own: the hot data kernel stays as it is).  dF: [REG_SLOTS][S][F],
// block b adds into copy b % REG_SLOTS; the reader sums the copies.
__global__ __launch_bounds__(REG_BLOCK) void reg_design_grad_kernel(RegView v, const float *__restrict__ p, const float *__restrict__ eps,
                                                                   const float *__restrict__ design, const float *__restrict__ W,
                                                                   const float *__restrict__ ss, const float *__restrict__ x,
                                                                   float *__restrict__ dF)
{
    __shared__ float s_weff[REG_MAXF][REG_BLOCK];
    const int tid = threadIdx.x;
    const int64_t jj = (int64_t)blockIdx.x * REG_BLOCK + tid;
    const bool live = jj < v.n;
    const int64_t j = live ? jj : v.n - 1;
    const int F = v.F, deg = v.deg, n = v.n;
    const int64_t Fn = v.Fn();
    for (int f = 0; f < F; ++f) {
        const int64_t idx = (int64_t)f * n + j;
        const float w = p[v.o_cols() + 8 * Fn + idx] + softplusf(p[v.o_cols() + 9 * Fn + idx]) * eps[v.e_cols() + 4 * Fn + idx];
        float wd = 0.0f;
        if (v.use_distortion)
            for (int d = 0; d < deg; ++d) wd += p[v.o_dist() + f * deg + d] * W[(int64_t)d * n + j];
        s_weff[f][tid] = w + wd;
    }
    const float b = p[v.o_bias_loc() + j] + softplusf(p[v.o_bias_s() + j]) * eps[v.e_bias() + j];
    const float xsz = softplusf(p[v.o_xs_loc() + j] + softplusf(p[v.o_xs_s() + j]) * eps[v.e_xs() + j]);
    const float inv = frcp(xsz), inv2 = inv * inv;
    float *out = dF + (int64_t)(blockIdx.x % REG_SLOTS) * v.S * F;
    for (int s = 0; s < v.S; ++s) {
        float xl = b;
        for (int f = 0; f < F; ++f) xl += design[s * F + f] * s_weff[f][tid];
        const int64_t sj = (int64_t)s * n + j;
        const float xv = v.point ? p[v.o_qx_loc() + sj] : x[sj];
        const float a = live ? (xv - (xl - ss[s])) * inv2 : 0.0f;
        for (int f = 0; f < F; ++f) {
            const float t = wave_sum(-a * s_weff[f][tid]);
            if ((tid & 63) == 0) atomicAdd(out + s * F + f, t);
        }
    }
}

}  // namespace polee

using namespace polee;

struct polee_regression {
    polee_ctx *ctx = nullptr;
    polee_approx *ap = nullptr;
    RegView v{};
    float lr = 2e-3f;
    int64_t step = 0;  // ADAM steps taken
    DevBuf<float> d_p, d_g, d_m, d_v, d_eps, d_design, d_W, d_ss, d_x, d_glik, d_lp, d_lse, d_small, d_loss, d_stats;
    DevBuf<float> d_lse_acc, d_lr, d_trace;
    DevBuf<uint32_t> d_tick;
    DevBuf<uint64_t> d_seed;
    hipGraphExec_t graph = nullptr;  // one step (device RNG), replayed by polee_regression_fit
    void drop_graph()
    {
        if (graph) (void)hipGraphExecDestroy(graph);
        graph = nullptr;
    }
    bool tick_in_finish = false;  // (per step) reg_finish_kernel advances the device clock: fit()'s steps
    bool x_drawn = false;         // (per step) reg_draw_kernel has written x: reg_data_pass skips reg_sample_x_kernel
    bool lse_valid = false;  // d_lse holds the log-sum-exp of a nearby qx_loc (shift of the multi-block kernel)
    DevBuf<float> d_lik_loc, d_lik_scale;  // point estimates + their scale: the Normal likelihood variant
    DevBuf<double> d_acc;
    // gene-level model: the likelihood handle over nt transcripts and the isoform block (see reg_iso_grad_kernel)
    polee_approx *gene_ap = nullptr;
    int32_t nt = 0;
    DevBuf<float> d_ip, d_ig, d_im, d_iv, d_ieps, d_xi;
    // gene-isoform model (RNASeqGeneIsoformLinearRegression, models/polee_regression.py:656-877): the isoform block
    // is a regression of its own over the transcripts -- horseshoe+ coefficients over the isoform design, x_isoform_bias
    // ~ Normal(0, 2), x_isoform_scale ~ InverseGamma(0.001, 0.001) -- laid out as a second RegView (vi, deg = 0) and
    // run through the same data / column / finish kernels
    bool iso_reg = false;
    // joint model (RNASeqJointLinearRegression): vi = the splice-feature block (P columns), the transcripts' part of the
    // block behind it in d_ip; the feature matrix both ways
    bool joint = false;
    int32_t P = 0;
    DevBuf<int32_t> d_tptr, d_tfeat, d_pptr, d_ptrans;
    DevBuf<float> d_mu, d_resid, d_hinges;
    std::vector<float> h_hinges;  // (kept from create: the joint model's weights follow the sampled bias)
    float bandwidth = 1.0f;
    RegView vi{};
    DevBuf<float> d_idesign, d_istats, d_ismall, d_iloss;
    DevBuf<double> d_iacc;
    int64_t num_iso_params() const
    {
        if (joint) return vi.num_params() + 2 * (int64_t)nt + 2 * (int64_t)v.S * nt;
        return !gene_ap ? 0 : (iso_reg ? vi.num_params() : 2 * (int64_t)nt + 2 * (int64_t)v.S * nt);
    }
    int64_t num_iso_noise() const
    {
        if (joint) return vi.num_noise() + (int64_t)nt + (int64_t)v.S * nt;
        return !gene_ap ? 0 : (iso_reg ? vi.num_noise() : (int64_t)nt + (int64_t)v.S * nt);
    }
    int64_t num_iso_stats() const { return (int64_t)(vi.F + 2) * vi.n + REG_SLOTS; }
    polee_comm *comm = nullptr;  // samples sharded over ranks: one all-reduce of d_stats per step
    int64_t num_stats() const { return (int64_t)(v.F + 2) * v.n + REG_SLOTS; }
    // classify (models/polee_regression.py:342-413): the design matrix is replaced per step (polee_regression_set_design), every
    // evaluation also leaves d loss / d design in d_dF, and Adam only moves the flat parameters [train_lo, train_hi)
    bool want_dgrad = false;
    DevBuf<float> d_dF;  // [REG_SLOTS][S][F]
    int64_t train_lo = 0, train_hi = -1;  // (-1: all of them)
};

namespace {

// data pass for the noise in d_eps: d_stats (this rank's samples) and the gradients of qx_*
polee_status reg_data_pass(polee_regression *r)
{
    polee_ctx *ctx = r->ctx;
    const RegView &v = r->v;
    hipStream_t st = ctx->stream;
    const int64_t sn = (int64_t)v.S * v.n;
    if (!v.point) {
        if (r->lse_valid) {
            hipLaunchKernelGGL(reg_lse_accum_kernel, dim3((unsigned)ceil_div(v.n, 4096), v.S), dim3(256), 0, st, v,
                               r->d_p.p, r->d_lse.p, r->d_lse_acc.p);
            hipLaunchKernelGGL(reg_lse_finish_kernel, dim3((unsigned)ceil_div(v.S, 64)), dim3(64), 0, st, v.S,
                               r->d_lse.p, r->d_lse_acc.p);
        } else
            hipLaunchKernelGGL(reg_lse_kernel, dim3(v.S), dim3(1024), 0, st, v, r->d_p.p, r->d_lse.p);
        r->lse_valid = true;
        if (!r->x_drawn)
            hipLaunchKernelGGL(reg_sample_x_kernel, dim3((unsigned)ceil_div(sn, 256)), dim3(256), 0, st, v, r->d_p.p,
                               r->d_eps.p, r->d_x.p);
        POLEE_KERNEL_CHECK(ctx);
        if (r->gene_ap) {
            if (r->joint)
                hipLaunchKernelGGL(reg_joint_sample_kernel, dim3((unsigned)ceil_div((int64_t)v.S * r->nt, 256)), dim3(256), 0,
                                   st, v.S, r->nt, (const float *)(r->d_ip.p + r->vi.num_params()),
                                   (const float *)(r->d_ieps.p + r->vi.num_noise()), r->d_xi.p);
            else if (r->iso_reg)
                hipLaunchKernelGGL(reg_sample_x_kernel, dim3((unsigned)ceil_div((int64_t)v.S * r->nt, 256)), dim3(256), 0,
                                   st, r->vi, r->d_ip.p, r->d_ieps.p, r->d_xi.p);
            else
                hipLaunchKernelGGL(reg_iso_sample_kernel, dim3((unsigned)ceil_div((int64_t)v.S * r->nt, 256)), dim3(256),
                                   0, st, v.S, r->nt, r->d_ip.p, r->d_ieps.p, r->d_xi.p);
            POLEE_KERNEL_CHECK(ctx);
            POLEE_TRY(approx_gene_logprob_device(r->gene_ap, r->d_x.p, r->d_xi.p, r->d_lp.p, r->d_glik.p));
        } else if (r->d_lik_loc.p) {
            hipLaunchKernelGGL(reg_normal_lik_kernel, dim3(v.S), dim3(1024), 0, st, v.n, r->d_x.p, r->d_lik_loc.p,
                               r->d_lik_scale.p, r->d_lp.p, r->d_glik.p);
            POLEE_KERNEL_CHECK(ctx);
        } else if (r->ap)
            POLEE_TRY(polee_approx_logprob_device(r->ap, r->d_x.p, r->d_lp.p, r->d_glik.p));
    }
    // (the loss slots of d_stats, d_acc and d_small are zero here: reg_finish_kernel clears what it reads)
    const float *lp = (!v.point && (r->ap || r->d_lik_loc.p || r->gene_ap)) ? r->d_lp.p : nullptr;
    const dim3 grid((unsigned)ceil_div(v.n, REG_BLOCK));
#define POLEE_REG_DATA(FT, DT)                                                                                          \
    hipLaunchKernelGGL((reg_data_kernel<FT, DT>), grid, dim3(REG_BLOCK), 0, st, v, r->d_p.p, r->d_eps.p, r->d_design.p, \
                       r->d_W.p, r->d_ss.p, r->d_x.p, r->d_glik.p, r->d_lse.p, lp, r->d_g.p, r->d_stats.p)
    // the reference's default degree (15) with up to four factors runs with fixed trip counts
    if (v.deg == 15 && v.F == 1) POLEE_REG_DATA(1, 15);
    else if (v.deg == 15 && v.F == 2) POLEE_REG_DATA(2, 15);
    else if (v.deg == 15 && v.F == 3) POLEE_REG_DATA(3, 15);
    else if (v.deg == 15 && v.F == 4) POLEE_REG_DATA(4, 15);
    else POLEE_REG_DATA(0, 0);
#undef POLEE_REG_DATA
    if (r->gene_ap && r->joint) {  // (d_xi now holds d lp / d x_iso): the splice block's predictor, the transcripts, back to the features
        const RegView &vs = r->vi;
        hipLaunchKernelGGL(reg_joint_mean_kernel, dim3((unsigned)ceil_div(vs.n, 128)), dim3(128), 0, st, vs, v.S, r->d_ip.p,
                           r->d_ieps.p, r->d_design.p, r->d_mu.p);
        hipLaunchKernelGGL(reg_joint_iso_kernel, dim3((unsigned)ceil_div(r->nt, 256)), dim3(256), 0, st, v.S, r->nt, r->P,
                           (const float *)(r->d_ip.p + vs.num_params()), (const float *)(r->d_ieps.p + vs.num_noise()),
                           (const float *)r->d_xi.p, (const float *)r->d_mu.p, r->d_tptr.p, r->d_tfeat.p,
                           r->d_ig.p + vs.num_params(), r->d_resid.p, r->d_stats.p + r->num_stats() - REG_SLOTS);
        hipLaunchKernelGGL(reg_joint_agg_kernel, dim3((unsigned)ceil_div(r->P, 128)), dim3(128), 0, st, v.S, r->nt, r->P, vs.F,
                           (const float *)r->d_resid.p, r->d_design.p, r->d_pptr.p, r->d_ptrans.p, r->d_istats.p);
    } else if (r->gene_ap && r->iso_reg)  // (d_xi now holds d lp / d x_isoform): the isoform block's own data pass
        hipLaunchKernelGGL((reg_data_kernel<0, 0>), dim3((unsigned)ceil_div(r->nt, REG_BLOCK)), dim3(REG_BLOCK), 0, st,
                           r->vi, r->d_ip.p, r->d_ieps.p, r->d_idesign.p, (const float *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, r->d_xi.p, (const float *)nullptr, (const float *)nullptr, r->d_ig.p,
                           r->d_istats.p);
    else if (r->gene_ap)
        hipLaunchKernelGGL(reg_iso_grad_kernel, dim3((unsigned)ceil_div(r->nt, 256)), dim3(256), 0, st, v.S, r->nt,
                           r->d_ip.p, r->d_ieps.p, r->d_xi.p, r->d_ig.p, r->d_stats.p + r->num_stats() - REG_SLOTS);
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

// prior pass: d_stats (summed over ranks) -> loss and the gradients of everything the ranks share
polee_status reg_prior_pass(polee_regression *r)
{
    polee_ctx *ctx = r->ctx;
    const RegView &v = r->v;
    hipStream_t st = ctx->stream;
    const dim3 grid((unsigned)ceil_div(v.n, REG_BLOCK));
#define POLEE_REG_COLS(FT, DT)                                                                                     \
    hipLaunchKernelGGL((reg_cols_kernel<FT, DT>), grid, dim3(REG_BLOCK), 0, st, v, r->d_p.p, r->d_eps.p, r->d_W.p, \
                       r->d_stats.p, r->d_g.p, r->d_acc.p, r->d_small.p)
    if (v.w_from_bias) POLEE_REG_COLS(0, 0);
    else if (v.deg == 15 && v.F == 1) POLEE_REG_COLS(1, 15);
    else if (v.deg == 15 && v.F == 2) POLEE_REG_COLS(2, 15);
    else if (v.deg == 15 && v.F == 3) POLEE_REG_COLS(3, 15);
    else if (v.deg == 15 && v.F == 4) POLEE_REG_COLS(4, 15);
    else POLEE_REG_COLS(0, 0);
#undef POLEE_REG_COLS
    if (r->iso_reg || r->joint) {  // the isoform / splice block's columns and its global scale; its loss joins the model's below
        hipLaunchKernelGGL((reg_cols_kernel<0, 0>), dim3((unsigned)ceil_div(r->vi.n, REG_BLOCK)), dim3(REG_BLOCK), 0, st,
                           r->vi, r->d_ip.p, r->d_ieps.p, (const float *)nullptr, r->d_istats.p, r->d_ig.p, r->d_iacc.p,
                           r->d_ismall.p);
        hipLaunchKernelGGL(reg_finish_kernel, dim3(1), dim3(64), 0, st, r->vi, r->d_ip.p, r->d_ieps.p, r->d_ismall.p,
                           r->d_istats.p, r->d_iacc.p, r->d_ig.p, r->d_iloss.p, (const float *)nullptr, (uint32_t *)nullptr, (float *)nullptr, 0.0f);
    }
    hipLaunchKernelGGL(reg_finish_kernel, dim3(1), dim3(64), 0, st, v, r->d_p.p, r->d_eps.p, r->d_small.p,
                       r->d_stats.p, r->d_acc.p, r->d_g.p, r->d_loss.p, (r->iso_reg || r->joint) ? r->d_iloss.p : (const float *)nullptr,
                       r->tick_in_finish ? r->d_tick.p : (uint32_t *)nullptr, r->d_lr.p, r->lr);
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

// loss and gradient at the current parameters for the noise in d_eps
polee_status reg_eval_device(polee_regression *r)
{
    POLEE_TRY(reg_data_pass(r));
    if (r->want_dgrad) {
        const RegView &v = r->v;
        POLEE_HIP_TRY(r->ctx, hipMemsetAsync(r->d_dF.p, 0, sizeof(float) * (size_t)REG_SLOTS * v.S * v.F, r->ctx->stream));
        hipLaunchKernelGGL(reg_design_grad_kernel, dim3((unsigned)ceil_div(v.n, REG_BLOCK)), dim3(REG_BLOCK), 0, r->ctx->stream, v,
                           r->d_p.p, r->d_eps.p, r->d_design.p, r->d_W.p, r->d_ss.p, r->d_x.p, r->d_dF.p);
        POLEE_KERNEL_CHECK(r->ctx);
    }
    if (r->comm && r->comm->nranks > 1)
        POLEE_TRY(comm_allreduce_device(r->comm, r->d_stats.p, (size_t)r->num_stats(), false));
    return reg_prior_pass(r);
}

// The latents every rank shares are drawn from (seed, step) alone, so that replicas stay identical; the noise of
// x belongs to a rank's own samples and is salted with the rank.
polee_status reg_fill_noise(polee_regression *r, const float *noise, uint64_t seed, uint32_t step, bool device_clock)
{
    polee_ctx *ctx = r->ctx;
    const int64_t ne = r->v.num_noise(), shared = r->v.e_x(), own = ne - shared;
    if (noise) {
        POLEE_TRY(r->d_eps.upload(ctx, noise, (size_t)ne));
        if (r->gene_ap) POLEE_TRY(r->d_ieps.upload(ctx, noise + ne, (size_t)r->num_iso_noise()));
        return POLEE_OK;
    }
    const uint64_t salt = 0xD1B54A32D192ED03ull * (uint64_t)(r->comm ? r->comm->rank + 1 : 1);
    const uint64_t *sd = device_clock ? r->d_seed.p : nullptr;
    const uint32_t *tk = device_clock ? r->d_tick.p : nullptr;
    hipLaunchKernelGGL(reg_noise_kernel, dim3((unsigned)ceil_div(ceil_div(shared, 4), 256)), dim3(256), 0, ctx->stream,
                       shared, seed, step, sd, tk, (uint64_t)0, r->d_eps.p);
    hipLaunchKernelGGL(reg_noise_kernel, dim3((unsigned)ceil_div(ceil_div(own, 4), 256)), dim3(256), 0, ctx->stream, own,
                       seed, step, sd, tk, salt, r->d_eps.p + shared);
    if (r->gene_ap)
        hipLaunchKernelGGL(reg_noise_kernel, dim3((unsigned)ceil_div(ceil_div(r->num_iso_noise(), 4), 256)), dim3(256), 0,
                           ctx->stream, r->num_iso_noise(), seed, step, sd, tk, salt ^ 0x69736f666f726d73ull, r->d_ieps.p);
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

// one step of fit() on the device clock: tick, draw, loss + gradient, Adam (+ the loss into the trace)
polee_status reg_enqueue_step(polee_regression *r, const float *noise, bool want_trace)
{
    polee_ctx *ctx = r->ctx;
    const int64_t P = r->v.num_params();
    if (noise) {  // the caller's noise: uploaded; the clock ticks in its own launch
        hipLaunchKernelGGL(reg_tick_kernel, dim3(1), dim3(1), 0, ctx->stream, r->d_tick.p, r->d_lr.p, r->lr);
        POLEE_TRY(reg_fill_noise(r, noise, 0, 0, true));
        POLEE_TRY(reg_eval_device(r));
    } else {
        // device RNG: one launch draws every latent of the step number the clock is ABOUT to show and x with it; the clock is
        // advanced by reg_finish_kernel (in front of Adam, behind everything that reads the noise)
        const int64_t ne = r->v.num_noise(), shared = r->v.e_x(), own = ne - shared;
        const uint64_t salt = 0xD1B54A32D192ED03ull * (uint64_t)(r->comm ? r->comm->rank + 1 : 1);
        const bool with_x = !r->v.point && own == (int64_t)r->v.S * r->v.n;
        hipLaunchKernelGGL(reg_draw_kernel, dim3((unsigned)ceil_div((shared + 3) / 4 + (own + 3) / 4, 256)), dim3(256), 0, ctx->stream, r->v,
                           shared, own, (const uint64_t *)r->d_seed.p, (const uint32_t *)r->d_tick.p, salt, (const float *)r->d_p.p, r->d_eps.p,
                           with_x ? r->d_x.p : (float *)nullptr);
        if (r->gene_ap)
            hipLaunchKernelGGL(reg_noise_kernel, dim3((unsigned)ceil_div(ceil_div(r->num_iso_noise(), 4), 256)), dim3(256), 0,
                               ctx->stream, r->num_iso_noise(), (uint64_t)0, 0u, (const uint64_t *)r->d_seed.p, (const uint32_t *)r->d_tick.p,
                               salt ^ 0x69736f666f726d73ull, r->d_ieps.p, 1u);
        r->tick_in_finish = true;
        r->x_drawn = with_x;
        const polee_status es = reg_eval_device(r);
        r->tick_in_finish = false;
        r->x_drawn = false;
        POLEE_TRY(es);
    }
    const int64_t lo = r->train_hi < 0 ? 0 : r->train_lo, cnt = r->train_hi < 0 ? P : r->train_hi - r->train_lo;
    hipLaunchKernelGGL(reg_adam_kernel, dim3((unsigned)std::max<int64_t>(ceil_div(cnt, 256), 1)), dim3(256), 0, ctx->stream, cnt,
                       r->d_p.p + lo, r->d_g.p + lo, r->d_m.p + lo, r->d_v.p + lo, r->d_lr.p, r->d_loss.p,
                       want_trace ? r->d_trace.p : nullptr, r->d_tick.p);
    if (r->gene_ap)
        hipLaunchKernelGGL(reg_adam_kernel, dim3((unsigned)ceil_div(r->num_iso_params(), 256)), dim3(256), 0, ctx->stream,
                           r->num_iso_params(), r->d_ip.p, r->d_ig.p, r->d_im.p, r->d_iv.p, r->d_lr.p, r->d_loss.p,
                           (float *)nullptr, r->d_tick.p);
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

}  // namespace

extern "C" {

polee_status polee_regression_create(polee_ctx *ctx, polee_approx *ap, int32_t S, int32_t F, int32_t n,
                                     const float *design, const float *x_init, const float *x_init_mean,
                                     const float *sample_scales, const float *hinges, int32_t degree, float bandwidth, float x_bias_loc0,
                                     float x_bias_scale0, int use_distortion, float scale_penalty,
                                     int use_point_estimates, polee_regression **out)
{
    POLEE_TRY(use_device(ctx));
    if (!out || !design || !x_init || !sample_scales || S < 1 || F < 1 || n < 1 || degree < 1)
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_regression_create: bad argument");
    if (F > REG_MAXF || degree > REG_MAXDEG)
        return fail(ctx, POLEE_ERR_UNSUPPORTED, "at most %d factors and %d hinges (got %d, %d)", REG_MAXF, REG_MAXDEG, F,
                    degree);
    if (!(bandwidth > 0.0f) || !(x_bias_scale0 > 0.0f) || (!use_point_estimates && !(scale_penalty > 0.0f)))
        return fail(ctx, POLEE_ERR_BAD_ARG, "bandwidth, x_bias_scale0 and scale_penalty must be positive");
    if (ap) {
        int32_t aS, an;
        approx_dims(ap, &aS, &an);
        if (approx_ctx(ap) != ctx || aS != S || an != n)
            return fail(ctx, POLEE_ERR_BAD_ARG, "the approximation handle holds %d x %d, the model %d x %d", aS, an, S, n);
    }
    polee_regression *r = new (std::nothrow) polee_regression();
    if (!r) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    r->ctx = ctx;
    ctx_retain(ctx);
    r->ap = use_point_estimates ? nullptr : ap;
    RegView &v = r->v;
    v = RegView{S, F, n, degree, use_distortion ? 1 : 0, use_point_estimates ? 1 : 0, x_bias_loc0, x_bias_scale0,
                use_point_estimates ? 1.0f : scale_penalty};
    const int64_t P = v.num_params(), sn = (int64_t)S * n, Fn = v.Fn();
    // initial values (models/polee_regression.py:49-119)
    std::vector<float> p((size_t)P, 0.0f);
    std::vector<double> mean((size_t)n, 0.0);
    if (x_init_mean)  // the column means over ALL samples when this handle holds a shard of them
        for (int j = 0; j < n; ++j) mean[(size_t)j] = x_init_mean[j];
    else {
        for (int s = 0; s < S; ++s)
            for (int j = 0; j < n; ++j) mean[(size_t)j] += x_init[(size_t)s * n + j];
        for (auto &m : mean) m /= S;
    }
    p[1] = p[3] = -1.0f;
    for (int d = 0; d < degree; ++d) p[(size_t)(v.o_conc() + d)] = p[(size_t)(v.o_scc() + d)] = 1.0f;
    for (int a = 1; a < 8; a += 2) std::fill_n(p.begin() + v.o_cols() + a * Fn, Fn, -1.0f);
    for (int j = 0; j < n; ++j) {
        p[(size_t)(v.o_bias_loc() + j)] = (float)mean[(size_t)j];
        p[(size_t)(v.o_bias_s() + j)] = -1.0f;
        p[(size_t)(v.o_xs_loc() + j)] = -0.5f;
        p[(size_t)(v.o_xs_s() + j)] = -1.0f;
    }
    std::copy_n(x_init, sn, p.begin() + v.o_qx_loc());
    std::fill_n(p.begin() + v.o_qx_s(), sn, -1.0f);
    // hinges (choose_knots, src/polee.py:69-76) and kernel-regression weights (:36-47)
    std::vector<double> hg((size_t)degree);
    if (hinges)
        for (int d = 0; d < degree; ++d) hg[(size_t)d] = hinges[d];
    else {
        const double lo = *std::min_element(mean.begin(), mean.end()), hi = *std::max_element(mean.begin(), mean.end());
        const double step = (hi - lo) / (degree + 1);
        for (int d = 0; d < degree; ++d) hg[(size_t)d] = lo + (d + 1) * step;
    }
    r->h_hinges.assign(hg.begin(), hg.end());
    r->bandwidth = bandwidth;
    std::vector<float> W((size_t)degree * n);
    for (int j = 0; j < n; ++j) {
        double tot = 0.0, col[REG_MAXDEG];
        for (int d = 0; d < degree; ++d) {
            const double u = ((double)(float)mean[(size_t)j] - hg[(size_t)d]) / bandwidth;
            col[d] = std::min(std::max(std::exp(-u * u), 1e-10), 1.0);
            tot += col[d];
        }
        for (int d = 0; d < degree; ++d) W[(size_t)d * n + j] = (float)(col[d] / tot);
    }
    polee_status st = POLEE_OK;
    auto ok = [&](polee_status s) { return st == POLEE_OK ? (st = s) == POLEE_OK : false; };
    if (ok(r->d_p.upload(ctx, p)) && ok(r->d_W.upload(ctx, W)) && ok(r->d_design.upload(ctx, design, (size_t)S * F)) &&
        ok(r->d_ss.upload(ctx, sample_scales, (size_t)S)) && ok(r->d_g.alloc(ctx, (size_t)P)) &&
        ok(r->d_m.alloc(ctx, (size_t)P)) && ok(r->d_v.alloc(ctx, (size_t)P)) &&
        ok(r->d_eps.alloc(ctx, (size_t)v.num_noise())) && ok(r->d_x.alloc(ctx, (size_t)sn)) &&
        ok(r->d_glik.alloc(ctx, (size_t)sn)) && ok(r->d_lp.alloc(ctx, (size_t)S)) && ok(r->d_lse.alloc(ctx, (size_t)S)) && ok(r->d_lse_acc.alloc(ctx, (size_t)S)) && ok(r->d_lr.alloc(ctx, 1)) && ok(r->d_tick.alloc(ctx, 2)) && ok(r->d_seed.alloc(ctx, 1)) &&
        ok(r->d_small.alloc(ctx, (size_t)REG_SLOTS * v.num_red())) && ok(r->d_stats.alloc(ctx, (size_t)r->num_stats())) && ok(r->d_loss.alloc(ctx, 1)) && ok(r->d_acc.alloc(ctx, REG_SLOTS))) {
        hipError_t e = hipMemsetAsync(r->d_m.p, 0, sizeof(float) * P, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(r->d_v.p, 0, sizeof(float) * P, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(r->d_glik.p, 0, sizeof(float) * sn, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(r->d_lse.p, 0, sizeof(float) * S, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(r->d_lse_acc.p, 0, sizeof(float) * S, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(r->d_acc.p, 0, sizeof(double) * REG_SLOTS, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(r->d_small.p, 0, sizeof(float) * REG_SLOTS * v.num_red(), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(r->d_stats.p, 0, sizeof(float) * r->num_stats(), ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = fail(ctx, POLEE_ERR_HIP, "memset failed: %s", hipGetErrorString(e));
    }
    if (st != POLEE_OK) {
        polee_regression_destroy(r);
        return st;
    }
    *out = r;
    return POLEE_OK;
}

void polee_regression_destroy(polee_regression *r)
{
    if (!r) return;
    polee_ctx *ctx = r->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    polee_comm_destroy(r->comm);
    r->drop_graph();
    delete r;
    ctx_release(ctx);
}

int64_t polee_regression_num_params(const polee_regression *r) { return r ? r->v.num_params() : 0; }
int64_t polee_regression_num_noise(const polee_regression *r) { return r ? r->v.num_noise() + r->num_iso_noise() : 0; }
int64_t polee_regression_num_isoform_params(const polee_regression *r) { return r ? r->num_iso_params() : 0; }
int64_t polee_debug_regression_num_stats(const polee_regression *r) { return r ? r->num_stats() : 0; }

polee_status polee_regression_get_params(polee_regression *r, float *params)
{
    if (!r || !params) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    POLEE_TRY(use_device(r->ctx));
    return r->d_p.download(r->ctx, params, (size_t)r->v.num_params());
}

polee_status polee_regression_set_params(polee_regression *r, const float *params)
{
    if (!r || !params) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    POLEE_TRY(use_device(r->ctx));
    r->lse_valid = false;
    return r->d_p.upload(r->ctx, params, (size_t)r->v.num_params());
}

polee_status polee_regression_weights(polee_regression *r, float *weights)
{
    if (!r || !weights) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    POLEE_TRY(use_device(r->ctx));
    return r->d_W.download(r->ctx, weights, (size_t)r->v.deg * r->v.n);
}

polee_status polee_regression_set_normal_likelihood(polee_regression *r, const float *loc, const float *scale)
{
    if (!r || !loc || !scale) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = r->ctx;
    POLEE_TRY(use_device(ctx));
    if (r->v.point) return fail(ctx, POLEE_ERR_BAD_ARG, "a model with point estimates has no likelihood term");
    const size_t sn = (size_t)r->v.S * r->v.n;
    for (size_t i = 0; i < sn; ++i)
        if (!(scale[i] > 0.0f)) return fail(ctx, POLEE_ERR_BAD_ARG, "scale[%zu] = %g is not positive", i, (double)scale[i]);
    r->drop_graph();
    POLEE_TRY(r->d_lik_loc.upload(ctx, loc, sn));
    return r->d_lik_scale.upload(ctx, scale, sn);
}

polee_status polee_regression_set_gene_likelihood(polee_regression *r, polee_approx *ap, const int32_t *gene_of,
                                                  const float *x_isoform_init)
{
    if (!r || !ap || !gene_of || !x_isoform_init) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = r->ctx;
    POLEE_TRY(use_device(ctx));
    int32_t aS, nt;
    approx_dims(ap, &aS, &nt);
    if (approx_ctx(ap) != ctx || aS != r->v.S)
        return fail(ctx, POLEE_ERR_BAD_ARG, "the approximation handle holds %d samples, the model %d", aS, r->v.S);
    if (r->v.point) return fail(ctx, POLEE_ERR_UNSUPPORTED, "the gene-level model is built without point estimates only");
    if (r->comm && r->comm->nranks > 1)
        return fail(ctx, POLEE_ERR_UNSUPPORTED, "the gene-level model is not sharded over ranks");
    POLEE_TRY(approx_set_genes(ap, gene_of, r->v.n));  // the model's features are the genes
    const int S = r->v.S;
    const size_t snt = (size_t)S * nt;
    std::vector<float> ip(2 * (size_t)nt + 2 * snt);
    for (int i = 0; i < nt; ++i) {  // models/polee_regression.py:573-577
        double m = 0.0;
        for (int s = 0; s < S; ++s) m += x_isoform_init[(size_t)s * nt + i];
        ip[(size_t)i] = (float)(m / S);
        ip[(size_t)nt + i] = -2.0f;
    }
    std::copy_n(x_isoform_init, snt, ip.begin() + 2 * (size_t)nt);
    std::fill_n(ip.begin() + 2 * (size_t)nt + snt, snt, -2.0f);
    r->drop_graph();
    r->gene_ap = ap;
    r->nt = nt;
    r->ap = nullptr;
    const size_t np = ip.size();
    POLEE_TRY(r->d_ip.upload(ctx, ip));
    POLEE_TRY(r->d_ig.alloc(ctx, np));
    POLEE_TRY(r->d_im.alloc(ctx, np));
    POLEE_TRY(r->d_iv.alloc(ctx, np));
    POLEE_TRY(r->d_ieps.alloc(ctx, (size_t)nt + snt));
    POLEE_TRY(r->d_xi.alloc(ctx, snt));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_im.p, 0, sizeof(float) * np, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_iv.p, 0, sizeof(float) * np, ctx->stream));
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return POLEE_OK;
}

polee_status polee_regression_set_gene_isoform_likelihood(polee_regression *r, polee_approx *ap, const int32_t *gene_of,
                                                          const float *x_isoform_init, const float *design_isoform,
                                                          int32_t num_isoform_factors)
{
    if (!r || !design_isoform) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = r->ctx;
    if (num_isoform_factors < 1 || num_isoform_factors > REG_MAXF)
        return fail(ctx, POLEE_ERR_UNSUPPORTED, "1..%d isoform factors (got %d)", REG_MAXF, num_isoform_factors);
    POLEE_TRY(polee_regression_set_gene_likelihood(r, ap, gene_of, x_isoform_init));
    const int S = r->v.S, nt = r->nt, Fi = num_isoform_factors;
    RegView &vi = r->vi;
    vi = RegView{S, Fi, nt, 0, 0, 0, 0.0f, 2.0f, 1.0f};
    vi.fixed_ab = 0.001f;
    const int64_t P = vi.num_params(), Fn = vi.Fn(), snt = (int64_t)S * nt;
    // initial values (models/polee_regression.py:740-775)
    std::vector<float> p((size_t)P, 0.0f);
    p[1] = p[3] = -1.0f;
    for (int a = 1; a < 10; a += 2) std::fill_n(p.begin() + vi.o_cols() + a * Fn, Fn, -1.0f);
    for (int i = 0; i < nt; ++i) {
        double m = 0.0;
        for (int s = 0; s < S; ++s) m += x_isoform_init[(size_t)s * nt + i];
        p[(size_t)(vi.o_bias_loc() + i)] = (float)(m / S);
        p[(size_t)(vi.o_bias_s() + i)] = -1.0f;
        p[(size_t)(vi.o_xs_loc() + i)] = 1.0f;
        p[(size_t)(vi.o_xs_s() + i)] = -1.0f;
    }
    std::copy_n(x_isoform_init, snt, p.begin() + vi.o_qx_loc());
    std::fill_n(p.begin() + vi.o_qx_s(), snt, -2.0f);
    r->iso_reg = true;
    POLEE_TRY(r->d_ip.upload(ctx, p));
    POLEE_TRY(r->d_idesign.upload(ctx, design_isoform, (size_t)S * Fi));
    POLEE_TRY(r->d_ig.alloc(ctx, (size_t)P));
    POLEE_TRY(r->d_im.alloc(ctx, (size_t)P));
    POLEE_TRY(r->d_iv.alloc(ctx, (size_t)P));
    POLEE_TRY(r->d_ieps.alloc(ctx, (size_t)vi.num_noise()));
    POLEE_TRY(r->d_istats.alloc(ctx, (size_t)r->num_iso_stats()));
    POLEE_TRY(r->d_ismall.alloc(ctx, (size_t)REG_SLOTS * vi.num_red()));
    POLEE_TRY(r->d_iacc.alloc(ctx, REG_SLOTS));
    POLEE_TRY(r->d_iloss.alloc(ctx, 1));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_im.p, 0, sizeof(float) * P, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_iv.p, 0, sizeof(float) * P, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_istats.p, 0, sizeof(float) * r->num_iso_stats(), ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_ismall.p, 0, sizeof(float) * REG_SLOTS * vi.num_red(), ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_iacc.p, 0, sizeof(double) * REG_SLOTS, ctx->stream));
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return POLEE_OK;
}

polee_status polee_regression_set_joint_likelihood(polee_regression *r, polee_approx *ap, const int32_t *gene_of,
                                                   const float *x_isoform_init, int32_t num_splice_features,
                                                   const int32_t *pair_transcript, const int32_t *pair_feature, int64_t num_pairs)
{
    if (!r || !pair_transcript || !pair_feature) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = r->ctx;
    if (num_splice_features < 1 || num_pairs < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "bad splice-feature matrix");
    if (r->v.use_distortion) return fail(ctx, POLEE_ERR_BAD_ARG, "the joint model has no distortion term: create the gene block with use_distortion = 0");
    POLEE_TRY(polee_regression_set_gene_likelihood(r, ap, gene_of, x_isoform_init));  // (gene_ap, nt, d_xi; its isoform block is replaced below)
    const int S = r->v.S, nt = r->nt, P = num_splice_features, F = r->v.F;
    for (int64_t q = 0; q < num_pairs; ++q)
        if (pair_transcript[q] < 0 || pair_transcript[q] >= nt || pair_feature[q] < 0 || pair_feature[q] >= P)
            return fail(ctx, POLEE_ERR_BAD_ARG, "splice-feature pair %lld out of range", (long long)q);
    // the feature matrix both ways (CSR by transcript, CSR by feature)
    std::vector<int32_t> tptr((size_t)nt + 1, 0), tfeat((size_t)num_pairs), pptr((size_t)P + 1, 0), ptrans((size_t)num_pairs);
    for (int64_t q = 0; q < num_pairs; ++q) {
        ++tptr[(size_t)pair_transcript[q] + 1];
        ++pptr[(size_t)pair_feature[q] + 1];
    }
    for (int i = 0; i < nt; ++i) tptr[(size_t)i + 1] += tptr[(size_t)i];
    for (int i = 0; i < P; ++i) pptr[(size_t)i + 1] += pptr[(size_t)i];
    {
        std::vector<int32_t> ct(tptr.begin(), tptr.end() - 1), cp(pptr.begin(), pptr.end() - 1);
        for (int64_t q = 0; q < num_pairs; ++q) {
            tfeat[(size_t)ct[(size_t)pair_transcript[q]]++] = pair_feature[q];
            ptrans[(size_t)cp[(size_t)pair_feature[q]]++] = pair_transcript[q];
        }
    }
    // the gene block becomes the joint model's: a horseshoe (one local level), weights from the sampled bias, HalfCauchy(0, 10)
    // on the mean-variance coefficients, qw_gene_softplus_scale = -2 (:1009-1041, :936-937), Adam(1e-3) (:1215)
    RegView &v = r->v;
    v.levels = 1;
    v.w_from_bias = 1;
    v.hc_scale = 10.0f;
    v.bandwidth = r->bandwidth;
    POLEE_TRY(r->d_hinges.upload(ctx, r->h_hinges));
    v.hinges = r->d_hinges.p;
    r->lr = 1e-3f;
    {
        std::vector<float> qs((size_t)v.Fn(), -2.0f);
        POLEE_HIP_TRY(ctx, hipMemcpy(r->d_p.p + v.o_cols() + 9 * v.Fn(), qs.data(), sizeof(float) * qs.size(), hipMemcpyHostToDevice));
    }
    // the splice block: P columns, horseshoe, bias ~ Normal(0, 10), no x_scale of its own (:1062-1087, surrogates :1172-1203)
    RegView &vs = r->vi;
    vs = RegView{0, F, P, 0, 0, 0, 0.0f, 10.0f, 1.0f};
    vs.levels = 1;
    vs.no_xs = 1;
    const int64_t PS = vs.num_params(), Fp = vs.Fn(), snt = (int64_t)S * nt, PJ = PS + 2 * (int64_t)nt + 2 * snt;
    std::vector<float> p((size_t)PJ, 0.0f);
    p[1] = p[3] = -1.0f;
    for (int a = 1; a < 8; a += 2) std::fill_n(p.begin() + vs.o_cols() + a * Fp, Fp, -1.0f);
    std::fill_n(p.begin() + vs.o_cols() + 9 * Fp, Fp, -2.0f);           // qw_splice_softplus_scale
    for (int i = 0; i < P; ++i) p[(size_t)(vs.o_bias_s() + i)] = -1.0f;  // qx_splice_bias: loc 0, softplus scale -1
    for (int i = 0; i < nt; ++i) {
        p[(size_t)(PS + i)] = 3.0f;        // qx_iso_scale_loc (:1004-1005)
        p[(size_t)(PS + nt + i)] = -1.0f;  // qx_iso_scale_softplus_scale
    }
    std::copy_n(x_isoform_init, snt, p.begin() + PS + 2 * (int64_t)nt);
    std::fill_n(p.begin() + PS + 2 * (int64_t)nt + snt, snt, -3.0f);  // qx_iso_softplus_scale (:1009-1010)
    r->joint = true;
    r->iso_reg = false;
    r->P = P;
    r->drop_graph();
    POLEE_TRY(r->d_ip.upload(ctx, p));
    POLEE_TRY(r->d_ig.alloc(ctx, (size_t)PJ));
    POLEE_TRY(r->d_im.alloc(ctx, (size_t)PJ));
    POLEE_TRY(r->d_iv.alloc(ctx, (size_t)PJ));
    POLEE_TRY(r->d_ieps.alloc(ctx, (size_t)r->num_iso_noise()));
    POLEE_TRY(r->d_istats.alloc(ctx, (size_t)r->num_iso_stats()));
    POLEE_TRY(r->d_ismall.alloc(ctx, (size_t)REG_SLOTS * vs.num_red()));
    POLEE_TRY(r->d_iacc.alloc(ctx, REG_SLOTS));
    POLEE_TRY(r->d_iloss.alloc(ctx, 1));
    POLEE_TRY(r->d_mu.alloc(ctx, (size_t)S * P));
    POLEE_TRY(r->d_resid.alloc(ctx, (size_t)snt));
    POLEE_TRY(r->d_tptr.upload(ctx, tptr));
    POLEE_TRY(r->d_pptr.upload(ctx, pptr));
    if (num_pairs > 0) {
        POLEE_TRY(r->d_tfeat.upload(ctx, tfeat));
        POLEE_TRY(r->d_ptrans.upload(ctx, ptrans));
    } else {
        POLEE_TRY(r->d_tfeat.alloc(ctx, 1));
        POLEE_TRY(r->d_ptrans.alloc(ctx, 1));
    }
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_ig.p, 0, sizeof(float) * PJ, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_im.p, 0, sizeof(float) * PJ, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_iv.p, 0, sizeof(float) * PJ, ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_istats.p, 0, sizeof(float) * r->num_iso_stats(), ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_ismall.p, 0, sizeof(float) * REG_SLOTS * vs.num_red(), ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(r->d_iacc.p, 0, sizeof(double) * REG_SLOTS, ctx->stream));
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return POLEE_OK;
}

polee_status polee_regression_set_learning_rate(polee_regression *r, float lr)
{
    if (!r || !(lr > 0.0f)) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "bad learning rate");
    r->lr = lr;
    r->drop_graph();  // (the step size is an argument of the captured tick kernel)
    return POLEE_OK;
}

polee_status polee_regression_get_isoform_params(polee_regression *r, float *params)
{
    if (!r || !params || !r->gene_ap) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "no isoform block");
    POLEE_TRY(use_device(r->ctx));
    return r->d_ip.download(r->ctx, params, (size_t)r->num_iso_params());
}

polee_status polee_regression_set_isoform_params(polee_regression *r, const float *params)
{
    if (!r || !params || !r->gene_ap) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "no isoform block");
    POLEE_TRY(use_device(r->ctx));
    return r->d_ip.upload(r->ctx, params, (size_t)r->num_iso_params());
}

// gradient of the isoform block left by the last polee_regression_eval
polee_status polee_regression_get_isoform_grad(polee_regression *r, float *grad)
{
    if (!r || !grad || !r->gene_ap) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "no isoform block");
    POLEE_TRY(use_device(r->ctx));
    return r->d_ig.download(r->ctx, grad, (size_t)r->num_iso_params());
}

polee_status polee_regression_set_comm(polee_regression *r, polee_comm *comm)
{
    if (!r) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    if (comm && comm->nranks > 1 && r->gene_ap)
        return fail(r->ctx, POLEE_ERR_UNSUPPORTED, "the gene-level model is not sharded over ranks");
    if (comm && comm->ctx != r->ctx)
        return fail(r->ctx, POLEE_ERR_BAD_ARG, "communicator and model belong to different contexts");
    if (comm) ++comm->refs;
    polee_comm_destroy(r->comm);
    r->comm = comm;
    r->drop_graph();
    return POLEE_OK;
}

// test hooks (include/polee_hip_debug.h): the two halves of a step, with the exchange left to the caller
polee_status polee_debug_regression_data_pass(polee_regression *r, const float *noise, float *stats)
{
    if (!r || !noise || !stats) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    POLEE_TRY(use_device(r->ctx));
    POLEE_TRY(reg_fill_noise(r, noise, 0, 0, false));
    POLEE_HIP_TRY(r->ctx, hipMemsetAsync(r->d_stats.p + r->num_stats() - REG_SLOTS, 0, sizeof(float) * REG_SLOTS,
                                         r->ctx->stream));
    POLEE_TRY(reg_data_pass(r));
    POLEE_TRY(r->d_stats.download(r->ctx, stats, (size_t)r->num_stats()));
    // no finish kernel follows a bare data pass: leave the loss slots clean for the next step
    POLEE_HIP_TRY(r->ctx, hipMemsetAsync(r->d_stats.p + r->num_stats() - REG_SLOTS, 0, sizeof(float) * REG_SLOTS,
                                         r->ctx->stream));
    return POLEE_OK;
}

polee_status polee_debug_regression_prior_pass(polee_regression *r, const float *stats, float *loss, float *grad)
{
    if (!r || !stats || !loss) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    POLEE_TRY(use_device(r->ctx));
    POLEE_TRY(r->d_stats.upload(r->ctx, stats, (size_t)r->num_stats()));
    POLEE_TRY(reg_prior_pass(r));
    POLEE_TRY(r->d_loss.download(r->ctx, loss, 1));
    if (grad) POLEE_TRY(r->d_g.download(r->ctx, grad, (size_t)r->v.num_params()));
    return POLEE_OK;
}

// ---- classify (models/polee_regression.py:342-413) ---------------------------------------------------------------------------------
polee_status polee_regression_set_design(polee_regression *r, const float *design)
{
    if (!r || !design) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = r->ctx;
    POLEE_TRY(use_device(ctx));
    if (r->gene_ap || r->v.fixed_ab > 0.0f || (r->comm && r->comm->nranks > 1))
        return fail(ctx, POLEE_ERR_UNSUPPORTED, "polee_regression_set_design: the transcript-level model on one GPU only");
    if (!r->want_dgrad) {
        POLEE_TRY(r->d_dF.alloc(ctx, (size_t)REG_SLOTS * r->v.S * r->v.F));
        r->want_dgrad = true;
        r->drop_graph();  // (the captured step gains a kernel)
    }
    return r->d_design.upload(ctx, design, (size_t)r->v.S * r->v.F);  // (same buffer: a captured step reads the new values)
}

polee_status polee_regression_set_trainable(polee_regression *r, int64_t begin, int64_t end)
{
    if (!r) return fail(nullptr, POLEE_ERR_BAD_ARG, "null argument");
    if (begin < 0 || end < begin || end > r->v.num_params()) return fail(r->ctx, POLEE_ERR_BAD_ARG, "polee_regression_set_trainable: [%lld, %lld) of %lld parameters", (long long)begin, (long long)end, (long long)r->v.num_params());
    r->train_lo = begin;
    r->train_hi = end;
    r->drop_graph();
    return POLEE_OK;
}

polee_status polee_regression_design_grad(polee_regression *r, float *grad)
{
    if (!r || !grad) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = r->ctx;
    POLEE_TRY(use_device(ctx));
    if (!r->want_dgrad) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_regression_design_grad: call polee_regression_set_design first");
    const size_t SF = (size_t)r->v.S * r->v.F;
    std::vector<float> h(SF * REG_SLOTS);
    POLEE_TRY(r->d_dF.download(ctx, h.data(), h.size()));
    for (size_t i = 0; i < SF; ++i) {
        double acc = 0.0;
        for (int c = 0; c < REG_SLOTS; ++c) acc += (double)h[(size_t)c * SF + i];
        grad[i] = (float)acc;
    }
    return POLEE_OK;
}

polee_status polee_regression_eval(polee_regression *r, const float *noise, uint64_t seed, float *loss, float *grad)
{
    if (!r || !loss) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "null argument");
    polee_ctx *ctx = r->ctx;
    POLEE_TRY(use_device(ctx));
    POLEE_TRY(reg_fill_noise(r, noise, seed, (uint32_t)(r->step + 1), false));
    POLEE_TRY(reg_eval_device(r));
    POLEE_TRY(r->d_loss.download(ctx, loss, 1));
    if (grad) POLEE_TRY(r->d_g.download(ctx, grad, (size_t)r->v.num_params()));
    return POLEE_OK;
}

polee_status polee_regression_fit(polee_regression *r, int32_t niter, uint64_t seed, const float *noise,
                                  float *loss_trace)
{
    if (!r || niter < 0) return fail(r ? r->ctx : nullptr, POLEE_ERR_BAD_ARG, "bad argument");
    polee_ctx *ctx = r->ctx;
    POLEE_TRY(use_device(ctx));
    if (niter == 0) return POLEE_OK;
    hipStream_t st = ctx->stream;
    const int64_t ne = r->v.num_noise() + r->num_iso_noise();
    if (r->d_trace.n < (size_t)niter) {  // (the trace's address is part of the captured step)
        r->drop_graph();
        POLEE_TRY(r->d_trace.alloc(ctx, std::max<size_t>((size_t)niter, 8192)));
    }
    const uint32_t clock[2] = {(uint32_t)r->step, 0u};
    POLEE_TRY(r->d_tick.upload(ctx, clock, 2));
    POLEE_TRY(r->d_seed.upload(ctx, &seed, 1));
    // Steps with the device RNG are replayed from a hipGraph (about 25 launches per step otherwise bound the step on
    // the host); supplied noise, a multi-rank communicator (RCCL inside a capture) or POLEE_REG_NO_GRAPH=1 enqueue
    // directly.  The first step always runs directly: it computes the exact log-sum-exp the later ones start from.
    const bool use_graph = !noise && !(r->comm && r->comm->nranks > 1) && !std::getenv("POLEE_REG_NO_GRAPH");
    int32_t it = 0;
    if (!use_graph || !r->lse_valid) {
        POLEE_TRY(reg_enqueue_step(r, noise, true));
        it = 1;
    }
    if (use_graph && it < niter && !r->graph) {
        hipGraph_t g = nullptr;
        POLEE_HIP_TRY(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const polee_status cs = reg_enqueue_step(r, nullptr, true);
        const hipError_t ce = hipStreamEndCapture(st, &g);
        if (cs != POLEE_OK || ce != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            return cs != POLEE_OK ? cs : fail(ctx, POLEE_ERR_HIP, "graph capture failed: %s", hipGetErrorString(ce));
        }
        const hipError_t ie = hipGraphInstantiate(&r->graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) {
            r->graph = nullptr;
            return fail(ctx, POLEE_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
        }
    }
    for (; it < niter; ++it) {
        if (use_graph)
            POLEE_HIP_TRY(ctx, hipGraphLaunch(r->graph, st));
        else
            POLEE_TRY(reg_enqueue_step(r, noise ? noise + (size_t)it * ne : nullptr, true));
    }
    r->step += niter;
    if (loss_trace) POLEE_TRY(r->d_trace.download(ctx, loss_trace, (size_t)niter));
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(st));
    return POLEE_OK;
}

}  // extern "C"
