// Density (and its gradient) of fitted likelihood approximations, batched over samples.
// Replaces RNASeqApproxLikelihoodDist._log_prob (src/polee_approx_likelihood.py:367-450)
// with its TF autodiff backward (InvHSB's registered gradient :17-28 -> InvHSBGrad,
// hsb_ops.cpp:342-391), and rnaseq_approx_likelihood_sampler (:35-59).
//
// Per sample s (row of the batch), with p = softmax(x), r = p * efflen, q = r / sum r:
//   leaf scan  : double-double prefix of q over the sample's DFS leaf order
//   nodes      : u_l, u_r from prefix differences; y = u_l/u; all per-node ladj / lp terms;
//                y_grad (for the VJP) and 1/u
//   (grad) tour scan : InvHSBGrad as an Euler-tour prefix sum of edge terms -> bp (d/dq)
//   (grad) finish    : chain through q = r/R, r = p*l, p = softmax(x):
//                      x_grad_j = q_j (bp_j - <bp,q> - 1) + 1 - (n-2) p_j
#include "ptt_internal.hpp"

#include <cmath>

namespace polee {

struct ApproxView {
    int32_t S, n;
    const float *efflens;              // [S][n]
    const float *mu, *sigma, *alpha;   // [S][n-1]
};

// per-sample sums: acc[s][0] = sum x, [1] = A = sum exp(x), [2] = Bn = sum exp(x)*efflen
__global__ void approx_sums_kernel(ApproxView a, const float *x, double *acc)
{
    __shared__ double smd[4];
    const int s = blockIdx.y;
    double sx = 0.0, A = 0.0, Bn = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.n; j += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[(int64_t)s * a.n + j];
        const double ex = (double)expf(xv);  // tf.math.exp(self.x) in f32 (:379)
        sx += (double)xv;
        A += ex;
        Bn += ex * (double)a.efflens[(int64_t)s * a.n + j];
    }
    sx = block_sum_f64(sx, smd);
    A = block_sum_f64(A, smd);
    Bn = block_sum_f64(Bn, smd);
    if (threadIdx.x == 0) {
        atomicAdd(&acc[s * 8 + 0], sx);
        atomicAdd(&acc[s * 8 + 1], A);
        atomicAdd(&acc[s * 8 + 2], Bn);
    }
}

// q (effective-length scaled, renormalised expression) of transcript tid of sample s
__device__ inline float approx_q(const ApproxView &a, const float *x, const double *acc, int s, int tid)
{
    const double ex = (double)expf(x[(int64_t)s * a.n + tid]);
    return (float)(ex * (double)a.efflens[(int64_t)s * a.n + tid] / acc[s * 8 + 2]);
}

struct ApproxLeafLoad {
    PttView v;
    ApproxView a;
    const float *x;
    const double *acc;
    __device__ dd operator()(int row, int64_t pos) const
    {
        const int tid = v.leaf_tid[(int64_t)v.tree(row) * v.n + pos];
        return dd_make((double)approx_q(a, x, acc, row, tid));
    }
};

// npart[s][block] = lp + ladj contributions of the block's internal nodes (summed by approx_finish_lp_kernel: one
// same-address atomic per block serialised ~800 deep per sample and dominated this kernel)
// tpair (optional) [S][n-1][2]: the node's two tour terms of InvHSBGrad, -1/u -+ ..., for its left ([0]) and right ([1])
// edge -- one 8-byte gather per tour element in the scan instead of three plus arithmetic
// 1 / x by the hardware estimate and two Newton steps (~1 ulp) for normal positive x; anything else (0, denormal, inf, nan) through the
// division, whose special cases the callers' tests pin
__device__ inline double approx_rcp(double x)
{
    if (!(x > 1e-300 && x < 1e300)) return 1.0 / x;
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// one internal node of sample s given the sums of q over its left / right subtree: lp + ladj contribution, and (tpair) its two tour terms
__device__ inline double approx_node_terms(const ApproxView &a, int s, int64_t k, int64_t nm1, double ul, double ur, double *tpair)
{
    // (The kernels that call this are bound by the instructions they issue -- 12.8 M nodes at S = 64 -- so: reciprocals by rcp +
    // Newton where the operand is a normal positive number, and the hyperbolics of :437-443 through their identities:
    // z = sinh(asinh(z_std) - alpha) = z_std cosh(alpha) - sqrt(1 + z_std^2) sinh(alpha), cosh(alpha - asinh(z_std)) = sqrt(1 + z^2).)
    const double u = ul + ur, iu = approx_rcp(u);
    const double y = ul * iu;                                   // hsb_ops.cpp:230
    double ladj = -fast_log(u);                                 // hsb_ops.cpp:231
    const double y_log = fast_log(y), y_1mlog = fast_log(1.0 - y);  // :418-419 (log1p(-y))
    const float y_logit = (float)(y_log - y_1mlog);             // :421
    ladj += (double)(float)(-y_log - y_1mlog);                  // :423-425
    const float muk = a.mu[(int64_t)s * nm1 + k], sg = a.sigma[(int64_t)s * nm1 + k];
    const float al = a.alpha[(int64_t)s * nm1 + k];
    const float isg = 1.0f / sg;
    const float z_std = (y_logit - muk) * isg;                  // :430
    ladj -= (double)logf(sg);                                   // :432
    const float root = sqrtf(fmaf(z_std, z_std, 1.0f));         // cosh(asinh(z_std))
    const float z = z_std * coshf(al) - root * sinhf(al);       // :437-438
    ladj += (double)(0.5f * log1pf(z * z) - 0.5f * log1pf(z_std * z_std));  // :440-443: log cosh(alpha - asinh(z_std)) = log sqrt(1 + z^2)
    const double lp = (-1.8378770664093453 - (double)(z * z)) / 2.0;              // :448, log(2 pi)
    if (tpair) {
        // d(lp + ladj)/d y_logit, then d y_logit / d y and the -log y - log1p(-y) term
        // with c = asinh(z_std) - alpha: sinh(c) = z, cosh(c) = sqrt(1 + z^2) (no f64 hyperbolics needed)
        const double zs = z_std, zd = (double)z, ch = sqrt(1.0 + zd * zd);
        const double rs = approx_rcp(sqrt(1.0 + zs * zs));
        const double d_lp = -zd * ch * rs;
        const double d_la = zd * approx_rcp(ch) * rs - zs * rs * rs;
        const double d_logit = (d_lp + d_la) * (double)isg;
        const double iy = approx_rcp(y), i1y = approx_rcp(1 - y);
        const double y_grad = d_logit * iy * i1y + (i1y - iy);
        double *tp = tpair + ((int64_t)s * nm1 + k) * 2;
        tp[0] = -iu + (-y) * iu * y_grad;
        tp[1] = -iu + (1.0 - y) * iu * y_grad;
    }
    return lp + ladj;
}

__global__ void approx_nodes_kernel(PttView v, ApproxView a, const dd *C, double *npart, double *tpair)
{
    __shared__ double smd[4];
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    const int64_t nm1 = v.n - 1;
    double contrib = 0.0;
    if (k < nm1) {
        const int64_t tb = (int64_t)v.tree(s) * nm1;
        const int lo = v.lo[tb + k], mid = v.mid[tb + k], hi1 = v.hi1[tb + k];
        const dd *Cr = C + (int64_t)s * (v.n + 1);
        const double ur = dd_diff(Cr[mid], Cr[lo]);
        const double ul = dd_diff(Cr[hi1], Cr[mid]);
        contrib = approx_node_terms(a, s, k, nm1, ul, ur, tpair);
    }
    contrib = block_sum_f64(contrib, smd);
    if (threadIdx.x == 0) npart[(int64_t)s * gridDim.x + blockIdx.x] = contrib;
}

// Round 6: the leaf prefix and the node terms in ONE launch per chunk of leaves (what vi_bwd_local_kernel does for the VI loop).
// A workgroup owns APX_CH leaf positions of sample s: it builds their chunk-LOCAL exclusive double-double prefix of q in LDS -- no
// offset from other chunks, so no reduce launch -- and, internal nodes in DFS pre-order having non-decreasing lo, evaluates every
// node whose leaf range starts AND ends in the chunk (node_start: the run of k that starts there) straight from LDS rows: the
// [S][n+1] prefix array (205 MB at S = 64, written once and gathered three times per node) never exists.  The nodes that cross
// chunks (a static list per tree) follow in approx_cross_nodes_kernel behind the spine, from the rows exported for them.
constexpr int APX_CH = 512;
struct ApproxChunkTables {
    const int32_t *node_start;  // [trees][nch + 1]
    const uint32_t *need;       // [trees][need_words]: bit pos = row pos is read by a crossing node
    const int32_t *cross;       // the trees' crossing nodes, concatenated
    const int32_t *cross_ptr;   // [trees + 1]
    int nch, need_words, max_cross_blocks;
};
__global__ __launch_bounds__(256) void approx_leaf_nodes_kernel(PttView v, ApproxView a, ApproxChunkTables T, const float *__restrict__ x,
                                                                const double *__restrict__ acc, dd *__restrict__ C /* [S][n+1], exported rows */,
                                                                dd *__restrict__ chunk_tot /* [S][nch+1] */, double *__restrict__ npart,
                                                                int npart_stride, double *tpair)
{
    __shared__ dd P[APX_CH + 1];
    __shared__ dd wtot[4];
    __shared__ double smd[4];
    const int s = blockIdx.y, tree = v.tree(s), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nm1 = v.n - 1, base = (int64_t)blockIdx.x * APX_CH, p0 = base + 2 * (int64_t)threadIdx.x;
    const int64_t end = min(base + APX_CH, (int64_t)v.n);
    const uint32_t *need = T.need + (size_t)tree * T.need_words;
    double q0 = 0.0, q1 = 0.0;
    if (p0 < v.n) q0 = (double)approx_q(a, x, acc, s, v.leaf_tid[(int64_t)tree * v.n + p0]);
    if (p0 + 1 < v.n) q1 = (double)approx_q(a, x, acc, s, v.leaf_tid[(int64_t)tree * v.n + p0 + 1]);
    const bool exp0 = p0 <= v.n && ((need[p0 >> 5] >> (p0 & 31)) & 1u);
    const bool exp1 = p0 + 1 <= v.n && ((need[(p0 + 1) >> 5] >> ((p0 + 1) & 31)) & 1u);
    const int32_t *ns = T.node_start + (size_t)tree * (T.nch + 1);
    const int32_t k0 = ns[blockIdx.x] + (int32_t)threadIdx.x, k1 = ns[blockIdx.x + 1];
    const dd sm{q0 + q1, (q0 - ((q0 + q1) - ((q0 + q1) - q0))) + (q1 - ((q0 + q1) - q0))};  // (TwoSum)
    const dd inc = wave_inclusive_scan<dd>(sm);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    dd off = ScanOps<dd>::shfl_up(inc, 1);
    if (lane == 0) off = dd{0.0, 0.0};
    for (int w = 0; w < wave; ++w) off = dd_add(wtot[w], off);
    const dd o1 = dd_add(off, dd_make(q0));
    P[2 * threadIdx.x] = off;
    P[2 * threadIdx.x + 1] = o1;
    dd *Cr = C + (int64_t)s * (v.n + 1);
    if (exp0) Cr[p0] = off;
    if (exp1) Cr[p0 + 1] = o1;
    if (threadIdx.x == 255) {
        const dd tot = dd_add(off, sm);
        P[APX_CH] = tot;
        chunk_tot[(int64_t)s * (T.nch + 1) + blockIdx.x] = tot;
        if (blockIdx.x == 0) chunk_tot[(int64_t)s * (T.nch + 1) + T.nch] = dd{0.0, 0.0};  // (the spine turns it into the sum over all leaves)
    }
    __syncthreads();
    double contrib = 0.0;
    const int64_t tb = (int64_t)tree * nm1;
    for (int32_t k = k0; k < k1; k += 256) {
        const int lo = v.lo[tb + k], mid = v.mid[tb + k], hi1 = v.hi1[tb + k];
        if (hi1 > end) continue;  // crosses chunks: approx_cross_nodes_kernel
        const dd pm = P[mid - base];
        contrib += approx_node_terms(a, s, k, nm1, dd_diff(P[hi1 - base], pm), dd_diff(pm, P[lo - base]), tpair);
    }
    contrib = block_sum_f64(contrib, smd);
    if (threadIdx.x == 0) npart[(int64_t)s * npart_stride + blockIdx.x] = contrib;
}
__global__ __launch_bounds__(256) void approx_cross_nodes_kernel(PttView v, ApproxView a, ApproxChunkTables T, const dd *__restrict__ C,
                                                                 const dd *__restrict__ chunk_off, double *__restrict__ npart,
                                                                 int npart_stride, double *tpair)
{
    __shared__ double smd[4];
    const int s = blockIdx.y, tree = v.tree(s);
    const int64_t nm1 = v.n - 1, tb = (int64_t)tree * nm1;
    const int32_t c0 = T.cross_ptr[tree], c1 = T.cross_ptr[tree + 1];
    const int32_t i = c0 + (int32_t)(blockIdx.x * 256 + threadIdx.x);
    double contrib = 0.0;
    if (i < c1) {
        const int32_t k = T.cross[i];
        const int lo = v.lo[tb + k], mid = v.mid[tb + k], hi1 = v.hi1[tb + k];
        const dd *Cr = C + (int64_t)s * (v.n + 1), *Or = chunk_off + (int64_t)s * (T.nch + 1);
        // (a boundary at a chunk's first position has the local prefix 0 by definition: its row is not exported -- row n of a tree
        // whose n is a multiple of the chunk size has no owner at all)
        auto G = [&](int b) { return b % APX_CH == 0 ? Or[b / APX_CH] : dd_add(Or[b / APX_CH], Cr[b]); };
        const dd gm = G(mid);
        const double ur = dd_diff(gm, G(lo));
        const double ul = dd_diff(G(hi1), gm);
        contrib = approx_node_terms(a, s, k, nm1, ul, ur, tpair);
    }
    contrib = block_sum_f64(contrib, smd);
    if (threadIdx.x == 0) npart[(int64_t)s * npart_stride + T.nch + blockIdx.x] = contrib;
}

// lp[s] = node terms + sum x - (n-1) log A + sum log efflen - log(R), R = Bn / A   (:384-400)
// (partials != nullptr: the gradient scan's per-chunk partial sums are finished here too -- dots[s][0..1], what a
// reduce_partials_kernel launch did: one single-workgroup launch per call less)
__global__ __launch_bounds__(256) void approx_finish_lp_kernel(ApproxView a, const double *acc, const double *npart,
                                                               int nblk, const double *sum_log_l, float *lp,
                                                               const double *partials, int nchunks, double *dots)
{
    __shared__ double smd[4];
    const int s = blockIdx.x;
    if (partials) {
        double s0 = 0.0, s1 = 0.0;
        for (int c = threadIdx.x; c < nchunks; c += 256) {
            s0 += partials[((int64_t)s * nchunks + c) * 2 + 0];
            s1 += partials[((int64_t)s * nchunks + c) * 2 + 1];
        }
        s0 = block_sum_f64(s0, smd);
        s1 = block_sum_f64(s1, smd);
        if (threadIdx.x == 0) {
            dots[(int64_t)s * 2 + 0] = s0;
            dots[(int64_t)s * 2 + 1] = s1;
        }
    }
    double nodes = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) nodes += npart[(int64_t)s * nblk + b];
    nodes = block_sum_f64(nodes, smd);
    if (threadIdx.x != 0) return;
    const double sx = acc[s * 8 + 0], A = acc[s * 8 + 1], Bn = acc[s * 8 + 2];
    const double ladj = sx - (double)(a.n - 1) * log(A) + sum_log_l[s] - log(Bn / A);
    lp[s] = (float)(nodes + ladj);
}

// InvHSBGrad (hsb_ops.cpp:342-391) with ladj_grad = 1 as an Euler-tour scan; 1/u_j comes
// straight from the leaf prefix because sum q = 1 (u_root = 1, as the op assumes).
struct ApproxGradLoad {
    PttView v;
    const double *tpair;  // [S][n-1][2], written by approx_nodes_kernel
    __device__ inline double term(int row, uint32_t code) const
    {
        if (code & 4u) return 0.0;
        return tpair[((int64_t)row * (v.n - 1) + (code >> 4)) * 2 + ((code >> 3) & 1u)];
    }
    __device__ dd operator()(int row, int64_t e) const
    {
        const uint32_t code = v.tour_code[(int64_t)v.tree(row) * v.TL + e];
        const uint32_t type = code & 3u;
        if (type == TOUR_LEAF) return dd_make(0.0);
        const double t = term(row, code);
        return dd_make(type == TOUR_ENTER ? t : -t);
    }
};
struct ApproxGradEmit {
    ApproxGradLoad l;
    ApproxView a;
    const float *x;
    const double *acc;
    float *bp;  // [S][n] by transcript
    __device__ void operator()(int row, int64_t e, dd /*excl*/, dd incl, double &p0, double &p1) const
    {
        p0 = p1 = 0.0;
        const int64_t tb = (int64_t)l.v.tree(row) * l.v.TL;
        const uint32_t code = l.v.tour_code[tb + e];
        if ((code & 3u) != TOUR_LEAF) return;
        const dd tot = dd_add(incl, dd_make(l.term(row, code)));
        const int pos = l.v.tour_tgt[tb + e];
        const int tid = l.v.leaf_tid[(int64_t)l.v.tree(row) * l.v.n + pos];
        const float b = (float)(tot.hi + tot.lo);  // backprops is float32 (hsb_ops.cpp:260)
        bp[(int64_t)row * l.v.n + tid] = b;
        p0 = (double)b * (double)approx_q(a, x, acc, row, tid);  // <bp, q>
    }
};

__global__ void approx_finish_grad_kernel(ApproxView a, const float *x, const double *acc, const double *dots,
                                          const float *bp, float *x_grad)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= a.n) return;
    const double p = (double)expf(x[(int64_t)s * a.n + j]) / acc[s * 8 + 1];
    const double q = (double)approx_q(a, x, acc, s, (int)j);
    const double dot = dots[s * 2];
    x_grad[(int64_t)s * a.n + j] =
        (float)(q * ((double)bp[(int64_t)s * a.n + j] - dot - 1.0) + 1.0 - (double)(a.n - 2) * p);
}

// sampler: z = sinh(asinh(z0) + alpha); y_logit = mu + sigma z; y = logistic (hsb_ops.cpp:103)
__global__ void approx_sample_y_kernel(ApproxView a, const float *z0, uint64_t seed, double *ys);
__global__ void approx_sample_finish_kernel(ApproxView a, const double *row_sums, float *x)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= a.n) return;
    // x_scaled = x_efflen / efflens; x = x_scaled / sum; clip (:54-58)
    float v = x[(int64_t)s * a.n + j] / a.efflens[(int64_t)s * a.n + j];
    v = (float)((double)v / row_sums[s * 2]);
    v = fminf(fmaxf(v, 1e-16f), 0.99999999f);
    x[(int64_t)s * a.n + j] = v;
}

__device__ inline float approx_philox_randn(uint64_t seed, uint32_t s, uint32_t k)
{
    // Philox4x32-10 as in vi.hip (kept local: separate translation unit)
    uint32_t c[4] = {k, s, 1u, 0x61707078u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ key[0], n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ key[1], n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        key[0] += 0x9E3779B9u;
        key[1] += 0xBB67AE85u;
    }
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__global__ void approx_sample_y_kernel(ApproxView a, const float *z0, uint64_t seed, double *ys)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    const int64_t nm1 = a.n - 1;
    if (k >= nm1) return;
    const int64_t o = (int64_t)s * nm1 + k;
    const float z0v = z0 ? z0[o] : approx_philox_randn(seed, (uint32_t)s, (uint32_t)k);
    const float z = sinhf(asinhf(z0v) + a.alpha[o]);
    const float y_logit = a.mu[o] + a.sigma[o] * z;
    ys[o] = 1.0 / (1.0 + (double)expf(-y_logit));
}

// ---- gene-level wrapper (polee_gene_expression.py:14-90) -------------------------------------------------
// x[s][i] = x_gene[s][g] + x_iso[s][i] - logsumexp_{i' in g} x_iso[s][i']; one thread per (sample, gene)
__global__ void gene_compose_kernel(const int32_t *gptr, const int32_t *gidx, int G, int n, const float *x_gene,
                                    const float *x_iso, float *x)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (g >= G) return;
    const float *xi = x_iso + (size_t)s * n;
    float mx = -INFINITY;
    for (int e = gptr[g]; e < gptr[g + 1]; ++e) mx = fmaxf(mx, xi[gidx[e]]);
    float sum = 0.0f;
    for (int e = gptr[g]; e < gptr[g + 1]; ++e) sum += expf(xi[gidx[e]] - mx);
    const float shift = x_gene[(size_t)s * G + g] - (mx + logf(sum));
    for (int e = gptr[g]; e < gptr[g + 1]; ++e) x[(size_t)s * n + gidx[e]] = xi[gidx[e]] + shift;
}
// VJP: gene_grad[g] = sum_{i in g} gx_i;  iso_grad[i] = gx_i - softmax_i * gene_grad[g]  (written over x_iso)
__global__ void gene_compose_grad_kernel(const int32_t *gptr, const int32_t *gidx, int G, int n, float *x_iso,
                                         const float *gx, float *gene_grad)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (g >= G) return;
    float *xi = x_iso + (size_t)s * n;
    const float *gxs = gx + (size_t)s * n;
    float mx = -INFINITY;
    for (int e = gptr[g]; e < gptr[g + 1]; ++e) mx = fmaxf(mx, xi[gidx[e]]);
    float sum = 0.0f, tot = 0.0f;
    for (int e = gptr[g]; e < gptr[g + 1]; ++e) {
        sum += expf(xi[gidx[e]] - mx);
        tot += gxs[gidx[e]];
    }
    gene_grad[(size_t)s * G + g] = tot;
    for (int e = gptr[g]; e < gptr[g + 1]; ++e) {
        const int i = gidx[e];
        xi[i] = gxs[i] - expf(xi[i] - mx) / sum * tot;
    }
}

}  // namespace polee

using namespace polee;

struct polee_approx {
    polee_ctx *ctx = nullptr;
    int32_t S = 0, n = 0;
    polee_ptt *t = nullptr;  // S trees (or 1 shared)
    DevBuf<float> d_efflens, d_mu, d_sigma, d_alpha;
    DevBuf<double> d_sum_log_l, d_acc, d_npart, d_tpair, d_dots;
    DevBuf<float> d_x, d_lp, d_xgrad, d_bp, d_z0;
    // the chunked leaf + node pass (approx_leaf_nodes_kernel): per tree the first node of every chunk's run, the prefix rows crossing
    // nodes read, the crossing nodes; the chunks' totals / offsets [S][nch + 1]
    DevBuf<int32_t> d_node_start, d_cross, d_cross_ptr;
    DevBuf<uint32_t> d_need;
    DevBuf<dd> d_chunk_tot;
    int nch = 0, need_words = 0, max_cross_blocks = 0;
    DevBuf<uint32_t> d_open_ptr, d_open_code;  // the trees' open-edge lists (build_open_lists): the gradient scan's chunk offsets
    DevBuf<int32_t> d_gptr, d_gidx;  // genes as CSR over transcripts (gene-level wrapper), set by approx_set_genes
    int32_t G = 0;
    ApproxView view() const { return ApproxView{S, n, d_efflens.p, d_mu.p, d_sigma.p, d_alpha.p}; }
};

namespace polee {
// genes as CSR (transcripts of gene g: gidx[gptr[g] .. gptr[g+1])), kept on the device
polee_status approx_set_genes(polee_approx *ap, const int32_t *gene_of, int32_t G)
{
    polee_ctx *ctx = ap->ctx;
    const int n = ap->n;
    if (!gene_of || G < 1 || G > n) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    std::vector<int32_t> gptr((size_t)G + 1, 0), gidx((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (gene_of[i] < 0 || gene_of[i] >= G)
            return fail(ctx, POLEE_ERR_BAD_ARG, "gene_of[%d] = %d outside 0..%d", i, gene_of[i], G - 1);
        ++gptr[(size_t)gene_of[i] + 1];
    }
    for (int g = 0; g < G; ++g) {
        if (gptr[(size_t)g + 1] == 0) return fail(ctx, POLEE_ERR_BAD_ARG, "gene %d has no transcript", g);
        gptr[(size_t)g + 1] += gptr[g];
    }
    {
        std::vector<int32_t> fill(gptr.begin(), gptr.end() - 1);
        for (int i = 0; i < n; ++i) gidx[(size_t)fill[gene_of[i]]++] = i;
    }
    POLEE_TRY(ap->d_gptr.upload(ctx, gptr));
    POLEE_TRY(ap->d_gidx.upload(ctx, gidx));
    ap->G = G;
    return POLEE_OK;
}

// RNASeqGeneApproxLikelihoodDist._log_prob on device buffers, enqueued on the context's stream: d_xg [S][G],
// d_xi [S][n] (OVERWRITTEN with d lp / d x_isoform when d_gg is given), d_lp [S], d_gg [S][G] or null.
polee_status approx_gene_logprob_device(polee_approx *ap, const float *d_xg, float *d_xi, float *d_lp, float *d_gg)
{
    polee_ctx *ctx = ap->ctx;
    if (ap->G < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "no gene map set");
    const int S = ap->S, n = ap->n, G = ap->G;
    const dim3 grid((unsigned)ceil_div(G, 256), (unsigned)S);
    hipLaunchKernelGGL(gene_compose_kernel, grid, dim3(256), 0, ctx->stream, ap->d_gptr.p, ap->d_gidx.p, G, n, d_xg,
                       d_xi, ap->d_x.p);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(polee_approx_logprob_device(ap, ap->d_x.p, d_lp, d_gg ? ap->d_xgrad.p : nullptr));
    if (d_gg) {
        hipLaunchKernelGGL(gene_compose_grad_kernel, grid, dim3(256), 0, ctx->stream, ap->d_gptr.p, ap->d_gidx.p, G, n,
                           d_xi, ap->d_xgrad.p, d_gg);
        POLEE_KERNEL_CHECK(ctx);
    }
    return POLEE_OK;
}

polee_ctx *approx_ctx(const polee_approx *ap) { return ap->ctx; }
void approx_dims(const polee_approx *ap, int32_t *S, int32_t *n)
{
    *S = ap->S;
    *n = ap->n;
}
}  // namespace polee

extern "C" {

polee_status polee_approx_create(polee_ctx *ctx, int32_t S, int32_t n, const float *efflens, const float *la_mu,
                                 const float *la_sigma, const float *la_alpha, const int32_t *left_index,
                                 const int32_t *right_index, const int32_t *leaf_index, int shared_tree,
                                 polee_approx **out)
{
    POLEE_TRY(use_device(ctx));
    if (!out || !efflens || !la_mu || !la_sigma || !la_alpha || !left_index || !right_index || !leaf_index || S < 1 ||
        n < 2)
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_approx_create: bad argument");
    polee_approx *ap = new (std::nothrow) polee_approx();
    if (!ap) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    ap->ctx = ctx;
    ctx_retain(ctx);
    ap->S = S;
    ap->n = n;
    const size_t sn = (size_t)S * n, sk = (size_t)S * (n - 1);
    std::vector<double> sll(S, 0.0);
    for (int32_t s = 0; s < S; ++s)
        for (int32_t j = 0; j < n; ++j) sll[s] += (double)logf(efflens[(size_t)s * n + j]);
    polee_status st = ptt_create_multi(ctx, left_index, right_index, leaf_index, shared_tree ? 1 : S, 2 * n - 1, &ap->t);
    auto A = [&](polee_status r) {
        if (st == POLEE_OK) st = r;
    };
    A(ap->d_efflens.upload(ctx, efflens, sn));
    A(ap->d_mu.upload(ctx, la_mu, sk));
    A(ap->d_sigma.upload(ctx, la_sigma, sk));
    A(ap->d_alpha.upload(ctx, la_alpha, sk));
    A(ap->d_sum_log_l.upload(ctx, sll));
    A(ap->d_acc.alloc(ctx, (size_t)S * 8));
    A(ap->d_npart.alloc(ctx, (size_t)S * (size_t)ceil_div(n - 1, 256)));
    A(ap->d_dots.alloc(ctx, (size_t)S * 2));
    A(ap->d_tpair.alloc(ctx, 2 * sk));
    A(ap->d_bp.alloc(ctx, sn));
    A(ap->d_x.alloc(ctx, sn));
    A(ap->d_xgrad.alloc(ctx, sn));
    A(ap->d_lp.alloc(ctx, S));
    if (st == POLEE_OK) st = ap->t->reserve(S);
    if (st == POLEE_OK && !getenv("POLEE_APPROX_NO_CHUNKS")) {  // (A/B: the three-phase leaf scan + approx_nodes_kernel)
        // tables of approx_leaf_nodes_kernel, per tree: internal nodes in DFS pre-order have non-decreasing lo
        const int nch = (int)ceil_div(n, APX_CH), need_words = (n + 1 + 31) / 32 + 1;
        std::vector<int32_t> node_start, cross, cross_ptr(1, 0);
        std::vector<uint32_t> need;
        bool ok = true;
        int max_cross = 0;
        for (const PttPlan &pl : ap->t->plans) {
            const size_t ns0 = node_start.size(), nd0 = need.size();
            node_start.resize(ns0 + (size_t)nch + 1, (int32_t)(n - 1));
            need.resize(nd0 + (size_t)need_words, 0u);
            int c = 0;
            for (int k = 0; k + 1 < n; ++k) {
                if (k > 0 && pl.lo[(size_t)k] < pl.lo[(size_t)k - 1]) ok = false;
                for (; c <= pl.lo[(size_t)k] / APX_CH; ++c) node_start[ns0 + (size_t)c] = k;
                const int64_t end = std::min<int64_t>(((int64_t)pl.lo[(size_t)k] / APX_CH + 1) * APX_CH, (int64_t)n);
                if (pl.hi1[(size_t)k] > end) {
                    cross.push_back(k);
                    for (int32_t b : {pl.lo[(size_t)k], pl.mid[(size_t)k], pl.hi1[(size_t)k]}) need[nd0 + ((size_t)b >> 5)] |= 1u << (b & 31);
                }
            }
            max_cross = std::max(max_cross, (int)cross.size() - cross_ptr.back());
            cross_ptr.push_back((int32_t)cross.size());
        }
        if (ok) {
            if (cross.empty()) cross.push_back(0);
            ap->nch = nch;
            ap->need_words = need_words;
            ap->max_cross_blocks = (int)ceil_div(max_cross, 256);
            A(ap->d_node_start.upload(ctx, node_start));
            A(ap->d_need.upload(ctx, need));
            A(ap->d_cross.upload(ctx, cross));
            A(ap->d_cross_ptr.upload(ctx, cross_ptr));
            A(ap->d_chunk_tot.alloc(ctx, (size_t)S * ((size_t)nch + 1)));
            A(ap->d_npart.alloc(ctx, (size_t)S * (size_t)std::max<int64_t>(ceil_div(n - 1, 256), (int64_t)nch + ap->max_cross_blocks)));
        }
    }
    if (st == POLEE_OK && !getenv("POLEE_APPROX_NO_OPEN_LISTS")) {  // (A/B)
        // the gradient scan's chunk offsets from the trees (build_open_lists): kept unless the trees are so deep that the lists
        // would pass 32 M entries in all -- the three-phase scan stays for those
        std::vector<uint32_t> optr, ocode;
        bool ok = true;
        for (const PttPlan &pl : ap->t->plans)
            ok = ok && (int64_t)pl.tour_code.size() == ap->t->TL &&
                 build_open_lists(pl.tour_code.data(), ap->t->TL, SCAN_CHUNK, (size_t)32 << 20, optr, ocode);
        if (ok) {
            if (ocode.empty()) ocode.push_back(4u | TOUR_LEAF);  // (never read)
            A(ap->d_open_ptr.upload(ctx, optr));
            A(ap->d_open_code.upload(ctx, ocode));
        }
    }
    if (st != POLEE_OK) {
        polee_approx_destroy(ap);
        return st;
    }
    *out = ap;
    return POLEE_OK;
}

void polee_approx_destroy(polee_approx *ap)
{
    if (!ap) return;
    polee_ctx *ctx = ap->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    polee_ptt_destroy(ap->t);
    delete ap;
    ctx_release(ctx);
}

polee_status polee_approx_logprob_device(polee_approx *ap, const float *d_x, float *d_lp, float *d_x_grad)
{
    if (!ap) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = ap->ctx;
    POLEE_TRY(use_device(ctx));
    if (!d_x || !d_lp) return fail(ctx, POLEE_ERR_BAD_ARG, "null argument");
    polee_ptt *t = ap->t;
    hipStream_t st = ctx->stream;
    const int S = ap->S, n = ap->n;
    const int64_t nm1 = n - 1;
    const bool grad = d_x_grad != nullptr;
    POLEE_HIP_TRY(ctx, hipMemsetAsync(ap->d_acc.p, 0, sizeof(double) * S * 8, st));
    // few blocks per sample: each ends in three same-address f64 atomics, which serialise (256 deep they cost ~20 us)
    const unsigned nb = (unsigned)std::min<int64_t>(ceil_div(n, 256), 40);
    hipLaunchKernelGGL(approx_sums_kernel, dim3(nb, S), dim3(256), 0, st, ap->view(), d_x, ap->d_acc.p);
    POLEE_KERNEL_CHECK(ctx);
    int nblk = (int)ceil_div(nm1, 256);
    if (ap->d_node_start.p) {
        const ApproxChunkTables T{ap->d_node_start.p, ap->d_need.p, ap->d_cross.p, ap->d_cross_ptr.p, ap->nch, ap->need_words, ap->max_cross_blocks};
        nblk = ap->nch + ap->max_cross_blocks;
        hipLaunchKernelGGL(approx_leaf_nodes_kernel, dim3((unsigned)ap->nch, S), dim3(256), 0, st, t->view(), ap->view(), T, d_x,
                           (const double *)ap->d_acc.p, t->d_C.p, ap->d_chunk_tot.p, ap->d_npart.p, nblk, grad ? ap->d_tpair.p : nullptr);
        if (ap->max_cross_blocks > 0) {
            hipLaunchKernelGGL((scan_spine_kernel<dd>), dim3(S), dim3(SCAN_THREADS), 0, st, ap->d_chunk_tot.p, ap->nch + 1);
            hipLaunchKernelGGL(approx_cross_nodes_kernel, dim3((unsigned)ap->max_cross_blocks, S), dim3(256), 0, st, t->view(), ap->view(), T,
                               (const dd *)t->d_C.p, (const dd *)ap->d_chunk_tot.p, ap->d_npart.p, nblk, grad ? ap->d_tpair.p : nullptr);
        }
    } else {
        ApproxLeafLoad load{t->view(), ap->view(), d_x, ap->d_acc.p};
        LeafPrefixEmit emit{n, t->d_C.p};
        hipError_t e0 = run_scan_partial<dd>(st, S, n, t->d_chunk.p, nullptr, load, emit);
        if (e0 != hipSuccess) return fail(ctx, POLEE_ERR_HIP, "scan launch failed: %s", hipGetErrorString(e0));
        hipLaunchKernelGGL(approx_nodes_kernel, dim3((unsigned)ceil_div(nm1, 256), S), dim3(256), 0, st, t->view(),
                           ap->view(), t->d_C.p, ap->d_npart.p, grad ? ap->d_tpair.p : nullptr);
    }
    hipError_t e = hipSuccess;
    if (!grad)
        hipLaunchKernelGGL(approx_finish_lp_kernel, dim3(S), dim3(256), 0, st, ap->view(), ap->d_acc.p, ap->d_npart.p,
                           nblk, ap->d_sum_log_l.p, d_lp, (const double *)nullptr, 0, (double *)nullptr);
    POLEE_KERNEL_CHECK(ctx);
    if (grad) {
        ApproxGradLoad gl{t->view(), ap->d_tpair.p};
        ApproxGradEmit ge{gl, ap->view(), d_x, ap->d_acc.p, ap->d_bp.p};
        if (ap->d_open_ptr.p) {
            // (chunk offsets from the trees' open-edge lists: the reduce launch and the spine of the three-phase scan are not needed)
            const int nch = scan_num_chunks(t->TL);
            hipLaunchKernelGGL((scan_apply_partial_open_kernel<ApproxGradLoad, ApproxGradEmit>), dim3(nch, S), dim3(SCAN_THREADS), 0, st, gl, ge,
                               t->TL, nch, (const uint32_t *)ap->d_open_ptr.p, (const uint32_t *)ap->d_open_code.p, t->T == 1 ? 0 : 1, t->d_part.p);
            e = hipGetLastError();
        } else {
            e = run_scan_partial<dd>(st, S, t->TL, t->d_chunk.p, t->d_part.p, gl, ge);
        }
        if (e != hipSuccess) return fail(ctx, POLEE_ERR_HIP, "scan launch failed: %s", hipGetErrorString(e));
        // (lp and the scan's two dot products in one single-workgroup launch per sample)
        hipLaunchKernelGGL(approx_finish_lp_kernel, dim3(S), dim3(256), 0, st, ap->view(), ap->d_acc.p, ap->d_npart.p,
                           nblk, ap->d_sum_log_l.p, d_lp, (const double *)t->d_part.p, scan_num_chunks(t->TL),
                           ap->d_dots.p);
        hipLaunchKernelGGL(approx_finish_grad_kernel, dim3((unsigned)ceil_div(n, 256), S), dim3(256), 0, st,
                           ap->view(), d_x, ap->d_acc.p, ap->d_dots.p, ap->d_bp.p, d_x_grad);
        POLEE_KERNEL_CHECK(ctx);
    }
    return POLEE_OK;
}

polee_status polee_approx_logprob(polee_approx *ap, const float *x, float *lp, float *x_grad)
{
    if (!ap) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = ap->ctx;
    POLEE_TRY(use_device(ctx));
    if (!x || !lp) return fail(ctx, POLEE_ERR_BAD_ARG, "null argument");
    const size_t sn = (size_t)ap->S * ap->n;
    POLEE_TRY(ap->d_x.upload(ctx, x, sn));
    POLEE_TRY(polee_approx_logprob_device(ap, ap->d_x.p, ap->d_lp.p, x_grad ? ap->d_xgrad.p : nullptr));
    POLEE_TRY(ap->d_lp.download(ctx, lp, ap->S));
    if (x_grad) POLEE_TRY(ap->d_xgrad.download(ctx, x_grad, sn));
    return POLEE_OK;
}

polee_status polee_approx_gene_logprob(polee_approx *ap, const float *x_gene, const float *x_isoform,
                                       const int32_t *gene_of, int32_t G, float *lp, float *gene_grad,
                                       float *isoform_grad)
{
    if (!ap) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = ap->ctx;
    POLEE_TRY(use_device(ctx));
    const int S = ap->S, n = ap->n;
    if (!x_gene || !x_isoform || !gene_of || !lp) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if ((gene_grad == nullptr) != (isoform_grad == nullptr))
        return fail(ctx, POLEE_ERR_BAD_ARG, "ask for both gradients or for none");
    POLEE_TRY(approx_set_genes(ap, gene_of, G));
    DevBuf<float> d_xg, d_xi, d_gg;
    const size_t sn = (size_t)S * n, sg = (size_t)S * G;
    POLEE_TRY(d_xg.upload(ctx, x_gene, sg));
    POLEE_TRY(d_xi.upload(ctx, x_isoform, sn));
    if (gene_grad) POLEE_TRY(d_gg.alloc(ctx, sg));
    POLEE_TRY(approx_gene_logprob_device(ap, d_xg.p, d_xi.p, ap->d_lp.p, gene_grad ? d_gg.p : nullptr));
    POLEE_TRY(ap->d_lp.download(ctx, lp, S));
    if (gene_grad) {
        POLEE_TRY(d_gg.download(ctx, gene_grad, sg));
        POLEE_TRY(d_xi.download(ctx, isoform_grad, sn));
    }
    return POLEE_OK;
}

// one draw per sample into ap->d_x (device side of polee_approx_sample); d_z0 [S][n-1] or null
static polee_status approx_sample_device(polee_approx *ap, const float *d_z0, uint64_t seed)
{
    polee_ctx *ctx = ap->ctx;
    polee_ptt *t = ap->t;
    const int S = ap->S, n = ap->n;
    hipLaunchKernelGGL(approx_sample_y_kernel, dim3((unsigned)ceil_div(n - 1, 256), S), dim3(256), 0, ctx->stream,
                       ap->view(), d_z0, seed, t->d_ys.p);
    POLEE_KERNEL_CHECK(ctx);
    FwdOut o;
    o.xs = ap->d_x.p;
    o.xs_rs = n;
    o.leaf_floor = 0.0;
    o.efflens = ap->d_efflens.p;
    o.efflens_rs = n;
    o.row_sums = ap->d_dots.p;
    POLEE_TRY(ptt_forward_device(t, t->d_ys.p, S, o));
    hipLaunchKernelGGL(approx_sample_finish_kernel, dim3((unsigned)ceil_div(n, 256), S), dim3(256), 0, ctx->stream,
                       ap->view(), ap->d_dots.p, ap->d_x.p);
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

polee_status polee_approx_sample(polee_approx *ap, const float *z0, uint64_t seed, float *x)
{
    if (!ap) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = ap->ctx;
    POLEE_TRY(use_device(ctx));
    if (!x) return fail(ctx, POLEE_ERR_BAD_ARG, "null argument");
    const size_t sn = (size_t)ap->S * ap->n, sk = (size_t)ap->S * (ap->n - 1);
    if (z0) POLEE_TRY(ap->d_z0.upload(ctx, z0, sk));
    POLEE_TRY(approx_sample_device(ap, z0 ? ap->d_z0.p : nullptr, seed));
    return ap->d_x.download(ctx, x, sn);
}

// ---- feature (gene) expression approximated by a normal, from sampler draws (polee_gene_expression.py:157-222)
__global__ void feature_sum_kernel(const int32_t *fidx, const int32_t *tidx, int64_t P, int n, int F, const float *x,
                                   float *fx)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (p >= P) return;
    atomicAdd(&fx[(size_t)s * F + fidx[p]], x[(size_t)s * n + tidx[p]]);
}
// mode 0: acc += log fx;  mode 1: acc += (loc - log fx)^2;  fx is reset for the next draw
__global__ void feature_accum_kernel(int64_t SF, float *fx, float *afx, const float *loc, double *acc, int mode)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SF) return;
    float v = logf(fx[i]);
    fx[i] = 0.0f;
    if (afx) {  // splicing log-ratio: log(feature) - log(antifeature) (polee_splicing.py:38-39)
        v -= logf(afx[i]);
        afx[i] = 0.0f;
    }
    if (mode == 0) {
        acc[i] += (double)v;
    } else {
        const float d = loc[i] - v;
        acc[i] += (double)(d * d);
    }
}
__global__ void feature_finish_kernel(int64_t SF, double *acc, double denom, int mode, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SF) return;
    out[i] = mode == 0 ? (float)(acc[i] / denom) : sqrtf((float)(acc[i] / denom));
    acc[i] = 0.0;
}

// shared driver: fi/ti (and optionally afi/ati) are 0-based device-ready pairs
static polee_status feature_moments_impl(polee_approx *ap, const std::vector<int32_t> &fi, const std::vector<int32_t> &ti,
                                         const std::vector<int32_t> &afi, const std::vector<int32_t> &ati, int32_t F,
                                         int32_t num_mean_draws, int32_t num_var_draws, uint64_t seed, const float *z0,
                                         float *loc, float *scale)
{
    polee_ctx *ctx = ap->ctx;
    const int S = ap->S, n = ap->n;
    const bool anti = !afi.empty();
    DevBuf<int32_t> d_fi, d_ti, d_afi, d_ati;
    DevBuf<float> d_fx, d_afx, d_loc, d_out;
    DevBuf<double> d_acc;
    const size_t SF = (size_t)S * F, sk = (size_t)S * (n - 1);
    POLEE_TRY(d_fi.upload(ctx, fi));
    POLEE_TRY(d_ti.upload(ctx, ti));
    POLEE_TRY(d_fx.alloc(ctx, SF));
    POLEE_TRY(d_loc.alloc(ctx, SF));
    POLEE_TRY(d_out.alloc(ctx, SF));
    POLEE_TRY(d_acc.alloc(ctx, SF));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d_fx.p, 0, SF * sizeof(float), ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d_acc.p, 0, SF * sizeof(double), ctx->stream));
    if (anti) {
        POLEE_TRY(d_afi.upload(ctx, afi));
        POLEE_TRY(d_ati.upload(ctx, ati));
        POLEE_TRY(d_afx.alloc(ctx, SF));
        POLEE_HIP_TRY(ctx, hipMemsetAsync(d_afx.p, 0, SF * sizeof(float), ctx->stream));
    }
    const int64_t P = (int64_t)fi.size(), Q = (int64_t)afi.size();
    const dim3 gp((unsigned)ceil_div(P, 256), (unsigned)S), gq((unsigned)ceil_div(std::max<int64_t>(Q, 1), 256), (unsigned)S),
        gf((unsigned)ceil_div((int64_t)SF, 256));
    int64_t draw = 0;
    for (int mode = 0; mode < 2; ++mode) {
        const int nd = mode == 0 ? num_mean_draws : num_var_draws;
        for (int d = 0; d < nd; ++d, ++draw) {
            if (z0) POLEE_TRY(ap->d_z0.upload(ctx, z0 + (size_t)draw * sk, sk));
            POLEE_TRY(approx_sample_device(ap, z0 ? ap->d_z0.p : nullptr, seed + 0x9E3779B97F4A7C15ull * (uint64_t)draw));
            hipLaunchKernelGGL(feature_sum_kernel, gp, dim3(256), 0, ctx->stream, d_fi.p, d_ti.p, P, n, F, ap->d_x.p,
                               d_fx.p);
            if (anti)
                hipLaunchKernelGGL(feature_sum_kernel, gq, dim3(256), 0, ctx->stream, d_afi.p, d_ati.p, Q, n, F,
                                   ap->d_x.p, d_afx.p);
            hipLaunchKernelGGL(feature_accum_kernel, gf, dim3(256), 0, ctx->stream, (int64_t)SF, d_fx.p,
                               anti ? d_afx.p : nullptr, d_loc.p, d_acc.p, mode);
        }
        hipLaunchKernelGGL(feature_finish_kernel, gf, dim3(256), 0, ctx->stream, (int64_t)SF, d_acc.p, (double)nd, mode,
                           mode == 0 ? d_loc.p : d_out.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    POLEE_TRY(d_loc.download(ctx, loc, SF));
    return d_out.download(ctx, scale, SF);
}

polee_status polee_approx_feature_moments(polee_approx *ap, const int32_t *feature_idxs, const int32_t *transcript_idxs,
                                          int64_t num_pairs, int32_t F, int32_t num_mean_draws, int32_t num_var_draws,
                                          uint64_t seed, const float *z0, float *loc, float *scale)
{
    if (!ap) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = ap->ctx;
    POLEE_TRY(use_device(ctx));
    const int n = ap->n;
    if (!feature_idxs || !transcript_idxs || !loc || !scale || num_pairs < 1 || F < 1 || num_mean_draws < 1 ||
        num_var_draws < 1)
        return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    std::vector<int32_t> fi((size_t)num_pairs), ti((size_t)num_pairs);
    for (int64_t p = 0; p < num_pairs; ++p) {  // 1-based, as the reference passes them
        if (feature_idxs[p] < 1 || feature_idxs[p] > F || transcript_idxs[p] < 1 || transcript_idxs[p] > n)
            return fail(ctx, POLEE_ERR_BAD_ARG, "pair %lld (%d, %d) out of range", (long long)p, feature_idxs[p],
                        transcript_idxs[p]);
        fi[(size_t)p] = feature_idxs[p] - 1;
        ti[(size_t)p] = transcript_idxs[p] - 1;
    }
    return feature_moments_impl(ap, fi, ti, {}, {}, F, num_mean_draws, num_var_draws, seed, z0, loc, scale);
}

polee_status polee_approx_splicing_moments(polee_approx *ap, const int32_t *feature_indices, int64_t num_feature_pairs,
                                           const int32_t *antifeature_indices, int64_t num_antifeature_pairs, int32_t F,
                                           int32_t num_mean_draws, int32_t num_var_draws, uint64_t seed, const float *z0,
                                           float *loc, float *scale)
{
    if (!ap) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = ap->ctx;
    POLEE_TRY(use_device(ctx));
    const int n = ap->n;
    if (!feature_indices || !antifeature_indices || !loc || !scale || num_feature_pairs < 1 || num_antifeature_pairs < 1 ||
        F < 1 || num_mean_draws < 1 || num_var_draws < 1)
        return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    auto unpack = [&](const int32_t *pairs, int64_t cnt, std::vector<int32_t> &f, std::vector<int32_t> &t) -> bool {
        f.resize((size_t)cnt);
        t.resize((size_t)cnt);
        for (int64_t p = 0; p < cnt; ++p) {  // rows (feature, transcript), 0-based (polee_splicing.py:69-80)
            f[(size_t)p] = pairs[2 * p];
            t[(size_t)p] = pairs[2 * p + 1];
            if (f[(size_t)p] < 0 || f[(size_t)p] >= F || t[(size_t)p] < 0 || t[(size_t)p] >= n) return false;
        }
        return true;
    };
    std::vector<int32_t> fi, ti, afi, ati;
    if (!unpack(feature_indices, num_feature_pairs, fi, ti) || !unpack(antifeature_indices, num_antifeature_pairs, afi, ati))
        return fail(ctx, POLEE_ERR_BAD_ARG, "feature / antifeature index out of range");
    return feature_moments_impl(ap, fi, ti, afi, ati, F, num_mean_draws, num_var_draws, seed, z0, loc, scale);
}

}  // extern "C"
