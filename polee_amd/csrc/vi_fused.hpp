// The K-vectorised tree pipeline of the VI loop.
//
// The generic tree kernels (ptt.hip) treat the K Monte-Carlo draws as K independent batch rows.  Inside the
// VI loop all K draws walk the SAME tree, so here one thread owns a tour entry / leaf / node for all K draws
// at once: the index arrays are read once instead of K times, the per-draw values sit next to each other
// ([...][K] layouts: one 48-byte access instead of six scattered 8-byte ones), and the K scan chains give
// each thread K-fold instruction-level parallelism.  Per VI iteration:
//   vi_sample      : z0 -> zs -> y, log y, log(1-y)                          (thread = node, K draws)
//   vi_fwd_reduce / scan_spine / vi_fwd_apply : Euler-tour scan -> leaf u, clamped x, zeroed g, sum x/efflen
//   [sparse likelihood pass]
//   vi_bwd_reduce / scan_spine / vi_bwd_apply : double-double leaf-order prefix of u*(g - efflen term)
//   vi_update      : y_grad, chain rule, mean over K, finiteness flag, ADAM  (thread = node)
#pragma once
#include "ptt_internal.hpp"

namespace polee {

// diagnostic build (-DPOLEE_VI_STAMPS, tools/probe/vi_stamps.py): thread 0 of every workgroup stamps the clock at the phase
// boundaries of the tree kernels (after waiting for the phase's memory operations, so the build is slower than the product)
#ifdef POLEE_VI_STAMPS
__device__ unsigned long long g_vi_stamps[4][2048][8];
#define VI_STAMP(kern, slot)                                                                                        \
    do {                                                                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 2048) g_vi_stamps[kern][blockIdx.x][slot] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define VI_STAMP(kern, slot) do {} while (0)
#endif

template <int K>
struct VK {
    double v[K];
};
template <int K>
struct VD {
    dd v[K];
};
template <int K>
struct ScanOps<VK<K>> {
    __device__ static VK<K> zero()
    {
        VK<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = 0.0;
        return r;
    }
    __device__ static VK<K> add(const VK<K> &a, const VK<K> &b)
    {
        VK<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = a.v[d] + b.v[d];
        return r;
    }
    __device__ static VK<K> shfl_up(const VK<K> &a, int dist)
    {
        VK<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = __shfl_up(a.v[d], dist, 64);
        return r;
    }
    template <int CTRL, int ROW_MASK>
    __device__ static VK<K> dpp(const VK<K> &a)
    {
        VK<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = dpp_f64<CTRL, ROW_MASK>(a.v[d]);
        return r;
    }
};
template <int K>
struct ScanOps<VD<K>> {
    __device__ static VD<K> zero()
    {
        VD<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = dd{0.0, 0.0};
        return r;
    }
    __device__ static VD<K> add(const VD<K> &a, const VD<K> &b)
    {
        VD<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = dd_add(a.v[d], b.v[d]);
        return r;
    }
    __device__ static VD<K> shfl_up(const VD<K> &a, int dist)
    {
        VD<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = dd{__shfl_up(a.v[d].hi, dist, 64), __shfl_up(a.v[d].lo, dist, 64)};
        return r;
    }
    template <int CTRL, int ROW_MASK>
    __device__ static VD<K> dpp(const VD<K> &a)
    {
        VD<K> r;
#pragma unroll
        for (int d = 0; d < K; ++d) r.v[d] = dd{dpp_f64<CTRL, ROW_MASK>(a.v[d].hi), dpp_f64<CTRL, ROW_MASK>(a.v[d].lo)};
        return r;
    }
};

// Block-wide sums of K doubles per thread (256 threads).  smem must hold 4*K doubles; results valid in
// threads 0..K-1 as out (and for every thread after the trailing barrier through smem[0..K-1]).
template <int K>
__device__ inline void block_sum_vec(double (&v)[K], double *smem)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < K; ++d) {
        const double s = wave_inclusive_scan<double>(v[d]);  // (DPP: the wave's sum lands in its last lane)
        if (lane == 63) smem[wave * K + d] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const double s = smem[threadIdx.x] + smem[K + threadIdx.x] + smem[2 * K + threadIdx.x] + smem[3 * K + threadIdx.x];
        smem[threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < K; ++d) v[d] = smem[d];
    __syncthreads();
}

// thread d stores v[d] to out[d] (no runtime-indexed register array)
template <int K>
__device__ inline void store_vec_by_thread(const double (&v)[K], double *out)
{
#pragma unroll
    for (int d = 0; d < K; ++d)
        if (threadIdx.x == d) out[d] = v[d];
}

// sinh(a + asinh(z)) = sinh(a) sqrt(1+z^2) + cosh(a) z : one sinh/cosh pair per node instead of an
// asinh + sinh per draw (src/sinh_arcsinh.jl:14-15)
__device__ inline float sinh_asinh(float sa, float ca, float z0) { return fmaf(sa, sqrtf(fmaf(z0, z0, 1.0f)), ca * z0); }

// sample: sinh_asinh_transform! (sinh_arcsinh.jl:10-23) -> logit_normal_transform! (logitnormal.jl:8-20).  What is kept per
// node and draw is the Float32 logistic value y32 [n-1][K] (4 bytes): the clamp of likelihood-approximation.jl:523 and the two
// edge logs log y / log(1 - y) are applied where the value is read (clamped_y, YRows) -- round 6: the f64 ys and the [n-1][2][K]
// f64 edge logs this kernel used to write were 29 MB of the iteration's 200 MB, and another 29 MB read back by the forward
// kernel.  ladj_out [K][2] = (skew, logit-normal) sums when non-null.
__device__ inline double clamped_y(float y32, double y_eps)
{
    const double y = (double)y32;
    return y < y_eps ? y_eps : (y > 1 - y_eps ? 1 - y_eps : y);
}
// (the same value when y_eps > 0, as the VI loop's is: the lower bound applied in f32 -- v_max against the largest float not
// above y_eps would round UP past values in (that float, y_eps), so the f64 compare stays for exactness but only one select
// each: written for the forward kernel, which clamps fourteen values per thread)
__device__ inline double clamped_y_pos(float y32, double y_eps, double one_m_eps)
{
    const double y = (double)y32;
    return fmin(fmax(y, y_eps), one_m_eps);
}
// one node's K draws (shared by the sample kernel and the update kernel's look-ahead)
template <int K, typename Noise>
__device__ inline void sample_node(float m, float om, float al, const Noise &noise, int step, int64_t k,
                                   float *__restrict__ y32, float *__restrict__ zcur, double *lsum)
{
    const float sigma = expf(om);
    const float sa = sinhf(al), ca = coshf(al);
    float zall[K];
    noise.template get_all<K>(step, k, zall);
#pragma unroll
    for (int d = 0; d < K; ++d) {
        const float z0 = zall[d];
        zcur[k * K + d] = z0;  // kept for this iteration's update (no second pass through the generator)
        const float zs = sinh_asinh(sa, ca, z0);
        const float yf = 1.0f / (1.0f + expf(-(m + zs * sigma)));
        if (lsum) {
            const double y = (double)yf;
            // log cosh(c) - 0.5 log1p(z0^2) with cosh(c) = sqrt(1 + sinh(c)^2)
            lsum[d] = 0.5 * (double)log1pf(zs * zs) - 0.5 * (double)log1pf(z0 * z0);
            lsum[K + d] = log((double)sigma * y * (1 - y));
        }
        y32[k * K + d] = yf;
    }
}

template <int K, typename Noise>
__global__ __launch_bounds__(256) void vi_sample_k_kernel(const float *__restrict__ mu, const float *__restrict__ omega,
                                                         const float *__restrict__ alpha, Noise noise, int step,
                                                         float *__restrict__ y32, float *__restrict__ zcur,
                                                         double *__restrict__ ladj_out)
{
    __shared__ double smem[4 * 2 * K];
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double lsum[2 * K];
#pragma unroll
    for (int d = 0; d < 2 * K; ++d) lsum[d] = 0.0;
    if (k < noise.nm1)
        sample_node<K, Noise>(mu[k], omega[k], alpha[k], noise, step, k, y32, zcur, ladj_out ? lsum : nullptr);
    if (ladj_out) {
        block_sum_vec<2 * K>(lsum, smem);
#pragma unroll
        for (int i = 0; i < 2 * K; ++i)
            if (threadIdx.x == i) atomicAdd(&ladj_out[(i % K) * 2 + i / K], lsum[i]);
    }
}

// Where a tour entry's edge log comes from: rows of precomputed logs (LogRows: lyy [n-1][2][K] f64, [0] = log(1-y) the right
// edge, [1] = log y the left edge -- the point optimisation), or the VI loop's Float32 y rows, clamped and logged here (YRows).
struct LogRows {
    const double *lyy;
    template <int K>
    __device__ inline double edge_one(uint32_t k, uint32_t left, int d) const { return lyy[((size_t)k * 2 + left) * K + d]; }
    template <int K>
    __device__ inline void edge(uint32_t k, uint32_t left, double (&e)[K]) const
    {
        const double *p = lyy + ((size_t)k * 2 + left) * K;
#pragma unroll
        for (int d = 0; d < K; ++d) e[d] = p[d];
    }
};
struct YRows {
    const float *y32;
    double y_eps;
    template <int K>
    __device__ inline double edge_one(uint32_t k, uint32_t left, int d) const
    {
        const double y = clamped_y_pos(y32[(size_t)k * K + d], y_eps, 1.0 - y_eps);
        return fast_log_pos(left ? y : 1.0 - y);
    }
    template <int K>
    __device__ inline void edge(uint32_t k, uint32_t left, double (&e)[K]) const
    {
        const float *p = y32 + (size_t)k * K;
        const double hi = 1.0 - y_eps;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            const double y = clamped_y_pos(p[d], y_eps, hi);
            e[d] = fast_log_pos(left ? y : 1.0 - y);  // (log1p(-y): y is clamped to [eps, 1 - eps], both arguments normal and positive)
        }
    }
};

// The VI loop's forward kernels take FWD_ITEMS tour entries per thread.  (Three per thread -- 782 workgroups at C2, all resident
// at four waves per SIMD where two per thread are 1 172 workgroups in two rounds -- was tried with the kernel held to 128
// VGPRs: 28.6 us against 29.0, and 9 spilled registers, whose scratch the runtime allocates lazily per queue.  Not kept.)
constexpr int FWD_ITEMS = 2;
constexpr int FWD_CHUNK = SCAN_THREADS * FWD_ITEMS;
inline int fwd_num_chunks(int64_t len) { return (int)((len + FWD_CHUNK - 1) / FWD_CHUNK); }

template <int K, typename Src>
__device__ inline VK<K> tour_value(uint32_t code, const Src &src, VK<K> &edge)
{
    // edge = log of this entry's edge factor (0 for the root); value = +edge on ENTER, -edge on EXIT, 0 on LEAF
    VK<K> val;
    const uint32_t type = code & 3u;
    if (code & 4u) {
#pragma unroll
        for (int d = 0; d < K; ++d) edge.v[d] = 0.0;
    } else {
        src.template edge<K>(code >> 4, (code >> 3) & 1u, edge.v);
    }
#pragma unroll
    for (int d = 0; d < K; ++d) val.v[d] = type == TOUR_ENTER ? edge.v[d] : (type == TOUR_EXIT ? -edge.v[d] : 0.0);
    return val;
}

// Sum of the totals of the chunks before `chunk` (its exclusive offset), computed by the workgroup itself: with a
// few hundred chunks this is cheaper than a separate single-workgroup spine launch between reduce and apply.
template <typename T>
__device__ inline T chunk_prefix(const T *__restrict__ chunk_sums, int chunk, T *smem)
{
    T acc = ScanOps<T>::zero();
    for (int i = threadIdx.x; i < chunk; i += SCAN_THREADS) acc = ScanOps<T>::add(acc, chunk_sums[i]);
    T tot;
    (void)block_exclusive_scan<T>(acc, smem, &tot);
    return tot;
}

template <int K, typename Src>
__global__ __launch_bounds__(SCAN_THREADS) void vi_fwd_reduce_kernel(PttView v, Src lyy,
                                                                    VK<K> *__restrict__ chunk_sums)
{
    __shared__ VK<K> smem[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * FWD_CHUNK + (int64_t)threadIdx.x * FWD_ITEMS;
    VK<K> acc = ScanOps<VK<K>>::zero(), edge;
#pragma unroll
    for (int j = 0; j < FWD_ITEMS; ++j)
        if (base + j < v.TL) acc = ScanOps<VK<K>>::add(acc, tour_value<K, Src>(v.tour_code[base + j], lyy, edge));
    VK<K> tot;
    (void)block_exclusive_scan<VK<K>>(acc, smem, &tot);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = tot;
}

// (UT: the type leaf u is kept in for the backward pass.  The VI loop keeps Float32 -- u multiplies an x gradient that comes out
// of f32 sums, the reference's own gradient intermediates are Float32 (ptt.jl:186-200 with T = Float32), and the f64 rows were
// 9.6 MB written and read back per iteration; the point optimisation keeps f64.)
// forward apply: leaves get u = exp(prefix + own edge); x = clamp(max(f32(u), 1e-16)) (ptt.jl:138-139,
// likelihood-approximation.jl:526) written to xs[tid][K]; g[tid][K] is zeroed for the likelihood pass;
// per-chunk partial sums of x/efflen (likelihood.jl:97-100) and, if wanted, of log u over internal nodes.
template <int K, typename Src, typename UT, bool LADJ>
__global__ __launch_bounds__(SCAN_THREADS) void vi_fwd_apply_kernel(PttView v, Src lyy,
                                                                   const VK<K> *__restrict__ chunk_offsets,
                                                                   UT *__restrict__ uleaf, float *__restrict__ xs,
                                                                   float *__restrict__ g,
                                                                   const float *__restrict__ efflens, float clamp_lo,
                                                                   float clamp_hi, double *__restrict__ part_c,
                                                                   double *__restrict__ part_ladj, int own_prefix,
                                                                   const uint32_t *__restrict__ open_ptr,
                                                                   const uint32_t *__restrict__ open_code,
                                                                   const uint32_t *__restrict__ tslot_ptr,
                                                                   const uint32_t *__restrict__ tslot,
                                                                   float *__restrict__ xwin,
                                                                   const float *__restrict__ single_cnt)
{
    __shared__ VK<K> smem[SCAN_THREADS / 64];
    __shared__ double smd[4 * K];
    __shared__ double spre[K];  // the chunk's offset from the open-edge list
    __shared__ double smd2[LADJ ? 4 * K : 1];
    // x windows of the sparse pass (loglik_internal.hpp): a transcript's x row also goes to its slot in every tile
    // dictionary that holds it (tslot lists) -- the gather launch in front of the pass is gone.  A thread writes up to
    // XW_INLINE slots itself; transcripts in more tiles are written by the whole workgroup afterwards.
    constexpr int XW_INLINE = 12, XW_DEFER = 48;
    __shared__ uint32_t xw_tid[XW_DEFER];
    __shared__ float xw_val[XW_DEFER][K];
    __shared__ int xw_count;
    if (xwin && threadIdx.x == 0) xw_count = 0;
    if (xwin) __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * FWD_CHUNK + (int64_t)threadIdx.x * FWD_ITEMS;
    // The kernel is bound by its dependent memory round trips, not by arithmetic: every load whose address does not depend on
    // another load is issued here, in front of the first barrier (the compiler does not move loads across one).
    //   trip 1: the thread's tour entries (code, target) and the bounds of the chunk's open-edge list
    //   trip 2: the entries' edge logs (lyy rows), the list's codes, the leaves' effective lengths / single counts
    //   trip 3: the open edges' logs
    VK<K> w[FWD_ITEMS];  // the entries' edge logs: +w on ENTER, -w on EXIT in the scan; a LEAF adds it to its prefix
    uint32_t code[FWD_ITEMS];
    int tgt[FWD_ITEMS];
#pragma unroll
    for (int j = 0; j < FWD_ITEMS; ++j) {
        code[j] = base + j < v.TL ? v.tour_code[base + j] : (4u | TOUR_LEAF);
        tgt[j] = base + j < v.TL ? v.tour_tgt[base + j] : 0;
    }
    VI_STAMP(0, 0);
    const int lane = threadIdx.x & 63;
    uint32_t ob = 0, oe = 0;
    if (open_ptr) {
        ob = open_ptr[blockIdx.x];
        oe = open_ptr[blockIdx.x + 1];
    }
    // (the list is the path from the root, rarely longer than a wave: every wave takes the entries' codes and sums the edge logs
    // of ITS draws -- wave w the draws w, w + 4 -- so the K double logs per entry are spread over the four waves; the sums go to
    // LDS in front of the scan's barriers)
    const int wave = threadIdx.x >> 6;
    const uint32_t ocode0 = ob + lane < oe ? open_code[ob + lane] : (4u | TOUR_LEAF);
    VK<K> acc = ScanOps<VK<K>>::zero();
    float inv_l_[FWD_ITEMS], sc_[FWD_ITEMS];
    int ltid[FWD_ITEMS];
#pragma unroll
    for (int j = 0; j < FWD_ITEMS; ++j) {
        acc = ScanOps<VK<K>>::add(acc, tour_value<K, Src>(code[j], lyy, w[j]));  // (past the end: a root LEAF, 0)
        const bool leaf = (code[j] & 3u) == TOUR_LEAF && base + j < v.TL;
        // (leaf-order mode, v.leaf_tid == null: the fit numbers the transcripts by leaf position, so xs, g, efflens
        // and single_cnt are indexed by pos and a chunk's leaves write one contiguous piece of each)
        ltid[j] = leaf ? (v.leaf_tid ? v.leaf_tid[tgt[j]] : tgt[j]) : 0;
        inv_l_[j] = leaf && efflens ? 1.0f / efflens[ltid[j]] : 0.0f;
        // stream S of the sparse pass (loglik_internal.hpp): the fragments compatible with this transcript alone
        // add cnt / x to its gradient -- g starts from that instead of 0
        sc_[j] = leaf && single_cnt ? single_cnt[ltid[j]] : 0.0f;
    }
    VI_STAMP(0, 1);
    if (open_ptr) {
        // The chunk's offset = the tour prefix in front of it.  Every ENTER before the chunk whose EXIT lies before it too has
        // cancelled, so the prefix is the sum of the edge logs of the nodes that are OPEN at the chunk's first entry -- the
        // path from the root to that point, a list fixed by the tree (open_ptr / open_code, built once per fit): with it the
        // forward pass needs no reduce launch and no pass over the other chunks' totals.
#pragma unroll
        for (int dd_ = 0; dd_ < (K + 3) / 4; ++dd_) {
            const int d = wave + 4 * dd_;  // (wave-uniform)
            if (d < K) {
                // (ENTER codes: + the edge's log; the filler is a root LEAF: 0)
                double pre = (ocode0 & 4u) ? 0.0 : lyy.template edge_one<K>(ocode0 >> 4, (ocode0 >> 3) & 1u, d);
                for (uint32_t e = ob + 64 + lane; e < oe; e += 64) {
                    const uint32_t oc = open_code[e];
                    pre += lyy.template edge_one<K>(oc >> 4, (oc >> 3) & 1u, d);
                }
                pre = wave_inclusive_scan<double>(pre);  // (DPP: the wave's sum lands in its last lane)
                if (lane == 63) spre[d] = pre;
            }
        }
    }
    VI_STAMP(0, 2);
    VK<K> tot;
    VK<K> off = block_exclusive_scan<VK<K>>(acc, smem, &tot);
    VI_STAMP(0, 3);
    // Without the lists (very deep trees): chunk_offsets holds the chunks' exclusive offsets (after a spine pass) or, with
    // own_prefix, their totals.
    if (open_ptr) {
#pragma unroll
        for (int d = 0; d < K; ++d) off.v[d] += spre[d];  // (written in front of the scan's barriers)
    } else {
        off = ScanOps<VK<K>>::add(own_prefix ? chunk_prefix<VK<K>>(chunk_offsets, blockIdx.x, smem) : chunk_offsets[blockIdx.x], off);
    }
    double pc[K], pl[K];
#pragma unroll
    for (int d = 0; d < K; ++d) pc[d] = pl[d] = 0.0;
#pragma unroll
    for (int j = 0; j < FWD_ITEMS; ++j) {
        const uint32_t tj = code[j] & 3u;
        // (the running prefix is advanced in place: + the edge's log on ENTER, - on EXIT, unchanged by a LEAF)
#pragma unroll
        for (int d = 0; d < K; ++d) off.v[d] = tj == TOUR_ENTER ? off.v[d] + w[j].v[d] : (tj == TOUR_EXIT ? off.v[d] - w[j].v[d] : off.v[d]);
        const VK<K> &inc = off;
        if (base + j < v.TL) {
            const uint32_t type = code[j] & 3u;
            if (type == TOUR_LEAF) {
                const int pos = tgt[j];
                const int tid = ltid[j];
                const float inv_l = inv_l_[j];
                const float sc = sc_[j];
#pragma unroll
                for (int d = 0; d < K; ++d) {
                    const double u = fast_exp(inc.v[d] + w[j].v[d]);
                    uleaf[(size_t)pos * K + d] = (UT)u;
                    float x = (float)u;
                    x = (float)fmax((double)x, 1e-16);
                    x = fminf(fmaxf(x, clamp_lo), clamp_hi);
                    xs[(size_t)tid * K + d] = x;
                    g[(size_t)tid * K + d] = sc != 0.0f ? sc / x : 0.0f;
                    pc[d] += (double)(x * inv_l);  // xls[i] = xs[i] / efflens[i] in f32 (likelihood.jl:97)
                }
                if (xwin) {
                    const uint32_t sb = tslot_ptr[tid], se = tslot_ptr[tid + 1];
                    int q = -1;
                    if (se - sb > (uint32_t)XW_INLINE) q = atomicAdd(&xw_count, 1);
                    if (q >= 0 && q < XW_DEFER) {
                        xw_tid[q] = (uint32_t)tid;
#pragma unroll
                        for (int d = 0; d < K; ++d) xw_val[q][d] = xs[(size_t)tid * K + d];
                    } else {
                        for (uint32_t e = sb; e < se; ++e) {
                            float *w = xwin + (size_t)tslot[e] * K;
#pragma unroll
                            for (int d = 0; d < K; ++d) w[d] = xs[(size_t)tid * K + d];
                        }
                    }
                }
            } else if (LADJ && type == TOUR_ENTER) {
#pragma unroll
                for (int d = 0; d < K; ++d) pl[d] += inc.v[d];
            }
        }
    }
    if (xwin) {
        __syncthreads();
        const int cnt = min(xw_count, XW_DEFER);
        for (int q = 0; q < cnt; ++q) {
            const uint32_t tid = xw_tid[q], sb = tslot_ptr[tid], se = tslot_ptr[tid + 1];
            for (uint32_t e = sb * K + threadIdx.x; e < se * K; e += SCAN_THREADS) {
                const uint32_t slot = e / K, d = e - slot * K;
                xwin[(size_t)tslot[slot] * K + d] = xw_val[q][d];
            }
        }
    }
    VI_STAMP(0, 4);
    // the chunk's partial sums: a wave's 64 values by a DPP scan, the four waves' through LDS in wave order -- one barrier
    if (part_c || LADJ) {
#pragma unroll
        for (int d = 0; d < K; ++d) {
            const double a = wave_inclusive_scan<double>(pc[d]);  // (DPP: the wave's sum lands in its last lane)
            const double b = LADJ ? wave_inclusive_scan<double>(pl[d]) : 0.0;
            if (lane == 63) {
                smd[wave * K + d] = a;
                if (LADJ) smd2[wave * K + d] = b;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < K) {
            const int d = threadIdx.x;
            if (part_c) part_c[(size_t)blockIdx.x * K + d] = ((smd[d] + smd[K + d]) + smd[2 * K + d]) + smd[3 * K + d];
            if (LADJ) part_ladj[(size_t)blockIdx.x * K + d] = ((smd2[d] + smd2[K + d]) + smd2[2 * K + d]) + smd2[3 * K + d];
        }
    }
    VI_STAMP(0, 5);
}

// gene_noninformative_prior! (likelihood.jl:114-159) inside the loop: gene_of == nullptr = off.
//   xl_grad_i = -(k_g - 1) / c_g for the k_g > 1 transcripts of gene g, c_g = sum of their xls (vi_gene_sums_kernel);
//   x_grad_i += xl_grad_i (1/efflen_i) / c + (1/efflen_i) offdiag / c^2,  c = sum_i x_i / efflen_i,
//   offdiag = sum_i -xl_grad_i xls_i = sum_g (k_g - 1) = M: a constant of the annotation (computed once on the host)
struct GenePrior {
    const int32_t *gene_of;
    const int *gene_k;     // transcripts per gene
    const double *gene_c;  // [num_genes][K]
    double M;
};

// summand of the backward scan at leaf position pos: a = u * x_grad, with the effective-length Jacobian
// term folded in (likelihood.jl:102-104): x_grad[j] -= n * (1/efflen_j) / sum_i x_i/efflen_i
template <int K>
__device__ inline void bwd_values(const PttView &v, int64_t pos, const double *__restrict__ uleaf,
                                  const float *__restrict__ g, const float *__restrict__ efflens,
                                  const double *csum, const GenePrior &gp, VD<K> &a)
{
    const int tid = v.leaf_tid ? v.leaf_tid[pos] : (int)pos;  // (leaf-order mode: see vi_fwd_apply_kernel)
    const float inv_lf = efflens ? 1.0f / efflens[tid] : 0.0f;
    const float nl = (float)v.n * inv_lf;  // Int * Float32 -> Float32 in the reference
    int gene = -1, kg = 0;
    if (gp.gene_of) {
        gene = gp.gene_of[tid];
        kg = gene >= 0 ? gp.gene_k[gene] : 0;
    }
#pragma unroll
    for (int d = 0; d < K; ++d) {
        double xg = (double)g[(size_t)tid * K + d];
        if (efflens) xg -= (double)nl / csum[d];
        if (gp.gene_of) {
            const double c = csum[d], inv_l = (double)inv_lf;
            const double xlg = kg > 1 ? -(double)(kg - 1) / gp.gene_c[(size_t)gene * K + d] : 0.0;
            xg += xlg * (inv_l / c) + inv_l * (gp.M / (c * c));
        }
        a.v[d] = dd_make(uleaf[(size_t)pos * K + d] * xg);
    }
}

// per-gene sums of xls = f32(f32(x / efflen) / c) for the prior above (likelihood.jl:126-128 on the xls that
// effective_length_jacobian_adjustment! leaves, likelihood.jl:97-100); gene_c must be zero on entry
template <int K>
__global__ __launch_bounds__(256) void vi_gene_sums_kernel(const float *__restrict__ xs, const float *__restrict__ efflens,
                                                          const double *__restrict__ part_c, int nchunks_fwd, int64_t n,
                                                          GenePrior gp, double *__restrict__ gene_c)
{
    __shared__ double smd[4 * K];
    double c[K];
#pragma unroll
    for (int d = 0; d < K; ++d) c[d] = 0.0;
    for (int ch = threadIdx.x; ch < nchunks_fwd; ch += 256)
#pragma unroll
        for (int d = 0; d < K; ++d) c[d] += part_c[(size_t)ch * K + d];
    block_sum_vec<K>(c, smd);
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= n) return;
    const int gene = gp.gene_of[tid];
    if (gene < 0 || gp.gene_k[gene] <= 1) return;
    const float inv_l = 1.0f / efflens[tid];
#pragma unroll
    for (int d = 0; d < K; ++d) {
        const float xl = xs[(size_t)tid * K + d] * inv_l;
        atomicAdd(&gene_c[(size_t)gene * K + d], (double)(float)((double)xl / c[d]));
    }
}

// backward reduce; its prologue also finishes sum x/efflen from the forward pass's per-chunk partials
template <int K>
__global__ __launch_bounds__(SCAN_THREADS) void vi_bwd_reduce_kernel(PttView v, const double *__restrict__ uleaf,
                                                                    const float *__restrict__ g,
                                                                    const float *__restrict__ efflens,
                                                                    const double *__restrict__ part_c, int nchunks_fwd,
                                                                    double *__restrict__ csum_out, GenePrior gp,
                                                                    VD<K> *__restrict__ chunk_sums)
{
    __shared__ VD<K> smem[SCAN_THREADS / 64];
    __shared__ double smd[4 * K];
    double c[K];
#pragma unroll
    for (int d = 0; d < K; ++d) c[d] = 0.0;
    if (efflens) {
        for (int ch = threadIdx.x; ch < nchunks_fwd; ch += SCAN_THREADS)
#pragma unroll
            for (int d = 0; d < K; ++d) c[d] += part_c[(size_t)ch * K + d];
        block_sum_vec<K>(c, smd);
        if (blockIdx.x == 0) store_vec_by_thread<K>(c, csum_out);
    }
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    VD<K> acc = ScanOps<VD<K>>::zero(), a;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < v.n) {
            bwd_values<K>(v, base + j, uleaf, g, efflens, c, gp, a);
            acc = ScanOps<VD<K>>::add(acc, a);
        }
    VD<K> tot;
    (void)block_exclusive_scan<VD<K>>(acc, smem, &tot);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = tot;
}

// backward apply: C[pos][K] = exclusive double-double prefix over leaf order, C[n] = total
template <int K>
__global__ __launch_bounds__(SCAN_THREADS) void vi_bwd_apply_kernel(PttView v, const double *__restrict__ uleaf,
                                                                   const float *__restrict__ g,
                                                                   const float *__restrict__ efflens,
                                                                   const double *__restrict__ csum, GenePrior gp,
                                                                   const VD<K> *__restrict__ chunk_offsets,
                                                                   dd *__restrict__ C, int own_prefix)
{
    __shared__ VD<K> smem[SCAN_THREADS / 64];
    double c[K];
#pragma unroll
    for (int d = 0; d < K; ++d) c[d] = efflens ? csum[d] : 1.0;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    VD<K> val[SCAN_ITEMS];
    VD<K> acc = ScanOps<VD<K>>::zero();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < v.n)
            bwd_values<K>(v, base + j, uleaf, g, efflens, c, gp, val[j]);
        else
            val[j] = ScanOps<VD<K>>::zero();
        acc = ScanOps<VD<K>>::add(acc, val[j]);
    }
    VD<K> tot;
    VD<K> off = block_exclusive_scan<VD<K>>(acc, smem, &tot);
    off = ScanOps<VD<K>>::add(own_prefix ? chunk_prefix<VD<K>>(chunk_offsets, blockIdx.x, smem) : chunk_offsets[blockIdx.x], off);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < v.n) {
#pragma unroll
            for (int d = 0; d < K; ++d) C[(size_t)(base + j) * K + d] = off.v[d];
        }
        off = ScanOps<VD<K>>::add(off, val[j]);
        if (base + j == v.n - 1) {
#pragma unroll
            for (int d = 0; d < K; ++d) C[(size_t)v.n * K + d] = off.v[d];
        }
    }
}

// 1 / x for a finite positive x well inside the normal range (the update kernel's denominators: y (1 - y) with y clamped to
// [y_eps, 1 - y_eps], sqrt(v) + eps): the hardware estimate and two Newton steps, ~1 ulp, a fifth of a division's instructions
__device__ inline double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

struct AdamConsts {
    double lr, rm, rv, eps, m_denom, v_denom;
    double inv_m_denom, inv_v_denom;  // reciprocals, computed once on the host (the update kernel is bound by its f64 instructions)
    double max_mu, max_omega, max_alpha;
    int first;  // step_num == 1
};

__device__ inline void adam_one(float &p, float &m, float &v, float grad, const AdamConsts &a, double max_step)
{
    // adam_update_mv! (likelihood-approximation.jl:116-130)
    if (a.first) {
        m = grad;
        v = grad * grad;
    } else {
        m = (float)(a.rm * (double)m + (1 - a.rm) * (double)grad);
        v = (float)(a.rv * (double)v + (1 - a.rv) * (double)(grad * grad));
    }
    // adam_update_params! (likelihood-approximation.jl:136-146) -- ascent, clamped step
    const double pm = (double)m * a.inv_m_denom, pv = (double)v * a.inv_v_denom;  // (m / (1 - rm^t) up to an ulp of double)
    double delta = a.lr * pm * fast_rcp(sqrt(pv) + a.eps);
    delta = delta < -max_step ? -max_step : (delta > max_step ? max_step : delta);
    p = (float)((double)p + delta);
}

// Everything the per-node update needs besides the node's two subtree sums.
struct UpdArgs {
    float *y32;  // [n-1][K] this iteration's logistic values (sample_node); the next iteration's after the update
    float *mu, *omega, *alpha, *m_mu, *v_mu, *m_omega, *v_omega, *m_alpha, *v_alpha;
    AdamConsts adam;
    int apply;
    int *nonfinite_step;
    double *y_grad_out;
    float *mu_grad_out, *omega_grad_out, *alpha_grad_out;
    int sample_next;
    double y_eps;
    float *zcur;
    int step;
};

// update of internal node k, all K draws.  hr[d] / hl[d] = sum of u * x_grad over the leaves of the node's right / left
// subtree for draw d (leaf positions [lo, mid) / [mid, hi1)).
//   y_grad[k] = H_l / y - H_r / (1 - y)   (closed form of ptt.jl:167-209, see ptt.hip)
//   logit_normal_transform_gradients! (logitnormal.jl:38-55),
//   sinh_asinh_transform_gradients! (sinh_arcsinh.jl:29-38) with cosh(c) = sqrt(1+zs^2), tanh(c) = zs/cosh(c),
//   omega_grad += sigma * sigma_grad (likelihood-approximation.jl:547-549), / K (:552-557), ADAM.
template <int K, typename Noise>
__device__ inline void update_node(const UpdArgs &A, const Noise &noise, int64_t k, int lo, int mid, int hi1, const double (&hr)[K],
                                   const double (&hl)[K])
{
    const int64_t nm1 = noise.nm1;
    const double cnt_r = (double)(mid - lo - 1), cnt_l = (double)(hi1 - mid - 1);
    const float muk = A.mu[k], omk = A.omega[k], alk = A.alpha[k];
    const float sigma = expf(omk), sa = sinhf(alk), ca = coshf(alk);
    float mu_g = 0.f, om_g = 0.f, al_g = 0.f;
    float p_mu = muk, p_om = omk, p_al = alk;
#pragma unroll
    for (int d = 0; d < K; ++d) {
        const double Hr = cnt_r + hr[d];
        const double Hl = cnt_l + hl[d];
        const double y = clamped_y(A.y32[k * K + d], A.y_eps);
        const double dyy = y * (1 - y);
        // H_l / y - H_r / (1 - y) over the common denominator, times its reciprocal (y is clamped: fast_rcp's range)
        const double ygd = (Hl * (1 - y) - Hr * y) * fast_rcp(dyy);
        if (A.y_grad_out) A.y_grad_out[(int64_t)d * nm1 + k] = ygd;
        const float yg = (float)ygd;  // y_grad is a Float32 array in the reference
        const float z0 = A.zcur[k * K + d];  // this iteration's draw, left by the sampling step
        const float zs = sinh_asinh(sa, ca, z0);
        const float cc = sqrtf(fmaf(zs, zs, 1.0f));  // cosh(alpha + asinh z0)
        mu_g = (float)((double)mu_g + dyy * (double)yg);  // mu_grad accumulates across draws in f32
        mu_g = (float)((double)mu_g + (1 - 2 * y));
        float sg = (float)(dyy * (double)zs * (double)yg);
        sg = (float)((double)sg + ((double)(1.0f / sigma) + (double)zs * (1 - 2 * y)));
        float zg = (float)(dyy * (double)sigma * (double)yg);
        zg = (float)((double)zg + (double)sigma * (1 - 2 * y));
        al_g += cc * zg;
        al_g += zs / cc;
        om_g += sigma * sg;
    }
    VI_STAMP(2, 2);
    mu_g /= (float)K;
    om_g /= (float)K;
    al_g /= (float)K;
    if (A.mu_grad_out) A.mu_grad_out[k] = mu_g;
    if (A.omega_grad_out) A.omega_grad_out[k] = om_g;
    if (A.alpha_grad_out) A.alpha_grad_out[k] = al_g;
    // (the gradient test hook, apply == 0, reports its gradients to the caller and must not leave a flag behind for a
    // step that was never applied)
    if (A.apply && !(isfinite(mu_g) && isfinite(om_g) && isfinite(al_g))) atomicCAS(A.nonfinite_step, 0, A.step);
    if (A.apply) {
        float p = muk, mm = A.m_mu[k], vv = A.v_mu[k];
        adam_one(p, mm, vv, mu_g, A.adam, A.adam.max_mu);
        A.mu[k] = p; A.m_mu[k] = mm; A.v_mu[k] = vv;
        p_mu = p;
        p = omk; mm = A.m_omega[k]; vv = A.v_omega[k];
        adam_one(p, mm, vv, om_g, A.adam, A.adam.max_omega);
        A.omega[k] = p; A.m_omega[k] = mm; A.v_omega[k] = vv;
        p_om = p;
        p = alk; mm = A.m_alpha[k]; vv = A.v_alpha[k];
        adam_one(p, mm, vv, al_g, A.adam, A.adam.max_alpha);
        A.alpha[k] = p; A.m_alpha[k] = mm; A.v_alpha[k] = vv;
        p_al = p;
    }
    VI_STAMP(2, 3);
    // look-ahead: the next iteration's draws from the parameters just written (this thread is the only one that
    // touches node k's y32 / zcur, and its reads of them are done)
    if (A.sample_next) sample_node<K, Noise>(p_mu, p_om, p_al, noise, A.step + 1, k, A.y32, A.zcur, nullptr);
}

// ---- backward + update, round 6 ----------------------------------------------------------------------------------------
// The backward pass used to be a GLOBAL double-double prefix over the leaves (a reduce launch for the chunk totals, an apply
// launch that re-read the leaves and added every chunk's offset, 19 MB of prefix rows) and the update gathered three rows of
// it per node.  Now:
//   vi_bwd_local_kernel : a workgroup owns a chunk of bu_ch(K) leaves.  It builds the chunk's LOCAL exclusive double-double
//                         prefix in LDS (no offset from the chunks before it: nothing to wait for, one read of the leaves).
//                         Internal nodes in DFS pre-order have non-decreasing lo, so the nodes whose leaf range STARTS in the
//                         chunk are one run of k (node_start): for every node of the run whose range also ENDS in the chunk
//                         the workgroup takes the two subtree sums as differences of LDS rows and writes them, draw-major,
//                         to H -- two doubles per node and draw, read back coalesced by the update.  Subtree sums inside a
//                         chunk never see a prefix that left the workgroup.
//   vi_bwd_spine_kernel : the chunks' totals -> exclusive chunk offsets (one workgroup).
//   vi_update_k_kernel  : a thread per node.  The few nodes whose range crosses a chunk boundary (about chunks x depth of them)
//                         take prefix(b) = off[chunk of b] + row b, from the rows the first kernel exported for them
//                         (need bits: a static set of the tree).
// Depth-independent like the scan it replaces: a caterpillar tree sends (almost) every node down the second path.
template <int K>
__host__ __device__ constexpr int bu_ch() { return K <= 6 ? 512 : 256; }  // leaves per workgroup (LDS: (ch + 1) K 16 B)

template <int K>
struct BwdArgs {
    PttView v;  // (leaf-order view)
    const float *uleaf;  // [n][K] leaf u, unclamped, as Float32 (the VI loop's forward kernel)
    const float *g, *efflens;
    const double *csum;  // [2][K]: sum x / efflen and its reciprocal (vi_csum_finish)
    GenePrior gp;
    VD<K> *chunk_tot;          // [nch + 1] the chunks' totals; the spine turns them into exclusive offsets, [nch] = all leaves
    dd *C;                     // [n + 1][K] chunk-local exclusive prefix rows, written only where need says so
    const uint32_t *need;      // bit pos: row pos is read by a node that crosses chunks
    const int32_t *node_start; // [nch + 1] first node k whose lo lies in chunk c
    float *H;                  // [2][K][n - 1] sums over the right / left subtree's leaves, nodes inside one chunk.  Float32:
                               // one rounding of a double-double difference -- the reference's own subtree gradients are
                               // Float32 sums of Float32 (ptt.jl:186-200, T = Float32), and y_grad is a Float32 array
};

// sums x / efflen over the forward pass's per-chunk partials: out[d] = sum, out[K + d] = 1 / sum.  Runs as one extra workgroup
// of the sparse pass's x-window gather (loglik.hip) or, without that launch, on its own.
template <int K>
__device__ inline void csum_finish_block(const double *__restrict__ part_c, int nparts, double *__restrict__ out, double *smem /* 4 K */)
{
    double c[K];
#pragma unroll
    for (int d = 0; d < K; ++d) c[d] = 0.0;
    for (int ch = threadIdx.x; ch < nparts; ch += 256)
#pragma unroll
        for (int d = 0; d < K; ++d) c[d] += part_c[(size_t)ch * K + d];
    block_sum_vec<K>(c, smem);
#pragma unroll
    for (int d = 0; d < K; ++d)
        if (threadIdx.x == d) {
            out[d] = c[d];
            out[K + d] = 1.0 / c[d];
        }
}
template <int K>
__global__ __launch_bounds__(256) void vi_csum_finish_kernel(const double *__restrict__ part_c, int nparts, double *__restrict__ out)
{
    __shared__ double smd[4 * K];
    csum_finish_block<K>(part_c, nparts, out, smd);
}

// a = u * x_grad at leaf position pos (bwd_values with the reciprocal of sum x / efflen at hand)
template <int K>
__device__ inline void bwd_leaf(const PttView &v, int64_t pos, const float *__restrict__ uleaf, const float *__restrict__ g,
                                const float *__restrict__ efflens, const double *c, const double *inv_c, const GenePrior &gp,
                                double (&a)[K])
{
    const int tid = v.leaf_tid ? v.leaf_tid[pos] : (int)pos;  // (leaf-order mode: see vi_fwd_apply_kernel)
    const float inv_lf = efflens ? 1.0f / efflens[tid] : 0.0f;
    const float nl = (float)v.n * inv_lf;  // Int * Float32 -> Float32 in the reference
    int gene = -1, kg = 0;
    if (gp.gene_of) {
        gene = gp.gene_of[tid];
        kg = gene >= 0 ? gp.gene_k[gene] : 0;
    }
#pragma unroll
    for (int d = 0; d < K; ++d) {
        double xg = (double)g[(size_t)tid * K + d];
        if (efflens) xg -= (double)nl * inv_c[d];
        if (gp.gene_of) {
            const double cc = c[d], inv_l = (double)inv_lf;
            const double xlg = kg > 1 ? -(double)(kg - 1) / gp.gene_c[(size_t)gene * K + d] : 0.0;
            xg += xlg * (inv_l / cc) + inv_l * (gp.M / (cc * cc));
        }
        a[d] = (double)uleaf[(size_t)pos * K + d] * xg;
    }
}

__device__ inline dd two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}

template <int K>
__global__ __launch_bounds__(256) void vi_bwd_local_kernel(BwdArgs<K> B)
{
    constexpr int CH = bu_ch<K>(), LPT = CH / 256;
    __shared__ dd P[(CH + 1) * K];
    __shared__ VD<K> wtot[4];
    const PttView &v = B.v;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double c[K], inv_c[K];
#pragma unroll
    for (int d = 0; d < K; ++d) {
        c[d] = B.efflens ? B.csum[d] : 1.0;
        inv_c[d] = B.efflens ? B.csum[K + d] : 1.0;
    }
    VI_STAMP(1, 0);
    const int64_t base = (int64_t)blockIdx.x * CH, p0 = base + (int64_t)threadIdx.x * LPT;
    const int64_t end = min(base + CH, (int64_t)v.n), nm1 = (int64_t)v.n - 1;
    const bool exp0 = p0 <= v.n && ((B.need[p0 >> 5] >> (p0 & 31)) & 1u);
    const bool exp1 = LPT == 2 && p0 + 1 <= v.n && ((B.need[(p0 + 1) >> 5] >> ((p0 + 1) & 31)) & 1u);
    double a[LPT][K];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        if (p0 + j < v.n)
            bwd_leaf<K>(v, p0 + j, B.uleaf, B.g, B.efflens, c, inv_c, B.gp, a[j]);
        else
#pragma unroll
            for (int d = 0; d < K; ++d) a[j][d] = 0.0;
    }
    // (the first nodes' ranges are loaded here, in front of the barriers -- the node phase then starts without a memory round
    // trip -- and BEHIND the leaves' loads: loads return in order, and these wait for node_start first)
    constexpr int PRE = LPT;
    // (node_start through the vector memory path -- a lane-dependent zero in the index: a scalar load would stall the wave
    // on its counter in front of the leaves' arithmetic)
    const uint32_t vz = threadIdx.x >> 12;
    const int32_t k0 = B.node_start[blockIdx.x + vz] + (int32_t)threadIdx.x, k1 = B.node_start[blockIdx.x + 1 + vz];
    int32_t kp[PRE];
    int plo[PRE], pmid[PRE], phi[PRE];
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        kp[i] = k0 + i * 256;
        const bool in = kp[i] < k1;
        plo[i] = in ? v.lo[kp[i]] : 0;
        pmid[i] = in ? v.mid[kp[i]] : 0;
        phi[i] = in ? v.hi1[kp[i]] : 0;
    }
    VD<K> s;
#pragma unroll
    for (int d = 0; d < K; ++d) {
        s.v[d] = dd_make(a[0][d]);
        if constexpr (LPT == 2) s.v[d] = two_sum(a[0][d], a[1][d]);
    }
    VI_STAMP(1, 1);
    const VD<K> inc = wave_inclusive_scan<VD<K>>(s);
    VI_STAMP(1, 2);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    VD<K> off = wave_shift_up_one<VD<K>>(inc);
    for (int w = 0; w < wave; ++w) off = ScanOps<VD<K>>::add(wtot[w], off);  // (wave-uniform trip count)
#pragma unroll
    for (int d = 0; d < K; ++d) P[(size_t)(threadIdx.x * LPT) * K + d] = off.v[d];
    if (exp0)
#pragma unroll
        for (int d = 0; d < K; ++d) B.C[(size_t)p0 * K + d] = off.v[d];
    if constexpr (LPT == 2) {
        const int64_t p1 = p0 + 1;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            const dd o1 = dd_add(off.v[d], dd_make(a[0][d]));
            P[(size_t)(threadIdx.x * LPT + 1) * K + d] = o1;
            if (exp1) B.C[(size_t)p1 * K + d] = o1;
        }
    }
    if (threadIdx.x == 255) {  // the chunk's total = the last thread's inclusive prefix
        VD<K> tot;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            tot.v[d] = dd_add(off.v[d], s.v[d]);
            P[(size_t)CH * K + d] = tot.v[d];
        }
        B.chunk_tot[blockIdx.x] = tot;
    }
    VI_STAMP(1, 3);
    __syncthreads();
    VI_STAMP(1, 4);
    // the nodes whose range starts in this chunk and ends in it: both subtree sums from LDS rows
    auto node = [&](int32_t k, int lo, int mid, int hi1) {
        if (hi1 > end) return;  // crosses chunks: the update adds chunk offsets to exported rows
        const dd *Plo = P + (size_t)(lo - base) * K, *Pmid = P + (size_t)(mid - base) * K, *Phi = P + (size_t)(hi1 - base) * K;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            const dd pm = Pmid[d];
            B.H[(size_t)d * nm1 + k] = (float)dd_diff(pm, Plo[d]);
            B.H[(size_t)(K + d) * nm1 + k] = (float)dd_diff(Phi[d], pm);
        }
    };
#pragma unroll
    for (int i = 0; i < PRE; ++i)
        if (kp[i] < k1) node(kp[i], plo[i], pmid[i], phi[i]);
    for (int32_t k = k0 + PRE * 256; k < k1; k += 256) node(k, v.lo[k], v.mid[k], v.hi1[k]);
    VI_STAMP(1, 5);
}

// exclusive double-double scan of the chunks' totals, in place; [nch] = the sum over all leaves.  One workgroup, one WAVE per
// draw: a lane adds its run of chunks, the wave scans the 64 runs, the lane walks its run again -- no barrier, no LDS, and a
// sixth of the dependent double-double chain a K-wide scan element has (this launch sits between the backward kernel and the
// update: its latency is the iteration's).
template <int K>
__global__ __launch_bounds__(64 * K) void vi_bwd_spine_kernel(VD<K> *chunk_tot, int nch)
{
    const int d = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (nch + 63) / 64;
    const int b = min(lane * per, nch), e = min(b + per, nch);
    constexpr int FAST = 8;  // runs of up to FAST chunks are held in registers: all loads in flight at once
    dd t[FAST];
    dd acc{0.0, 0.0};
    if (per <= FAST) {
#pragma unroll
        for (int i = 0; i < FAST; ++i) t[i] = b + i < e ? chunk_tot[b + i].v[d] : dd{0.0, 0.0};
#pragma unroll
        for (int i = 0; i < FAST; ++i) acc = dd_add(acc, t[i]);
    } else {
        for (int i = b; i < e; ++i) acc = dd_add(acc, chunk_tot[i].v[d]);
    }
    const dd inc = wave_inclusive_scan<dd>(acc);
    dd off = ScanOps<dd>::shfl_up(inc, 1);
    if (lane == 0) off = dd{0.0, 0.0};
    if (per <= FAST) {
#pragma unroll
        for (int i = 0; i < FAST; ++i) {
            if (b + i < e) chunk_tot[b + i].v[d] = off;
            off = dd_add(off, t[i]);
        }
    } else {
        for (int i = b; i < e; ++i) {
            const dd ti = chunk_tot[i].v[d];
            chunk_tot[i].v[d] = off;
            off = dd_add(off, ti);
        }
    }
    if (lane == 63) chunk_tot[nch].v[d] = inc;
}

// update: one thread per internal node k, all K draws
template <int K, typename Noise>
__global__ __launch_bounds__(256) void vi_update_k_kernel(PttView v, const float *__restrict__ H, const VD<K> *__restrict__ chunk_off,
                                                         const dd *__restrict__ C, UpdArgs A, Noise noise)
{
    constexpr int CH = bu_ch<K>();
    VI_STAMP(2, 0);
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nm1 = (int64_t)v.n - 1;
    if (k >= nm1) return;
    const int lo = v.lo[k], mid = v.mid[k], hi1 = v.hi1[k];
    double hr[K], hl[K];
    if ((int64_t)hi1 <= min(((int64_t)lo / CH + 1) * CH, (int64_t)v.n)) {
#pragma unroll
        for (int d = 0; d < K; ++d) {
            hr[d] = (double)H[(size_t)d * nm1 + k];
            hl[d] = (double)H[(size_t)(K + d) * nm1 + k];
        }
    } else {  // crosses chunks: a boundary's global prefix = its chunk's offset + its chunk-local row
        const dd *Plo = C + (size_t)lo * K, *Pmid = C + (size_t)mid * K, *Phi = C + (size_t)hi1 * K;
        const dd *Olo = chunk_off[lo / CH].v, *Omid = chunk_off[mid / CH].v, *Ohi = chunk_off[hi1 / CH].v;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            const dd gm = dd_add(Omid[d], Pmid[d]);
            hr[d] = dd_diff(gm, dd_add(Olo[d], Plo[d]));
            hl[d] = dd_diff(dd_add(Ohi[d], Phi[d]), gm);
        }
    }
    VI_STAMP(2, 1);
    update_node<K, Noise>(A, noise, k, lo, mid, hi1, hr, hl);
    VI_STAMP(2, 6);
}

// ---- point optimisation (OptimizePTTApprox, likelihood-approximation.jl:149-242), K = 1 ------------------
// ys = logistic(zs) without clamp (:196-198)
__global__ void point_sample_kernel(const float *__restrict__ zs, int64_t nm1, double *__restrict__ ys,
                                    double *__restrict__ lyy)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nm1) return;
    double y = (double)(1.0f / (1.0f + expf(-zs[k])));
    // the f32 logistic saturates at exactly 0 or 1 (the reference does not clamp here); its tree recursion
    // stays finite there, the closed form H_l/y - H_r/(1-y) needs the limit: keep y one ulp inside (0, 1) --
    // with the double-double leaf prefix the subtree sums stay exact at that scale
    y = y < 0x1p-52 ? 0x1p-52 : (y > 1 - 0x1p-52 ? 1 - 0x1p-52 : y);
    ys[k] = y;
    lyy[k * 2 + 0] = log1p(-y);
    lyy[k * 2 + 1] = log(y);
}
// z_grad = y (1-y) y_grad with transform_gradients_no_ladj! (ptt.jl:217-251; :208-213), ADAM on z (:226-227)
__global__ void point_update_kernel(PttView v, const double *__restrict__ ys, const dd *__restrict__ C, float *zs,
                                    float *m_z, float *v_z, AdamConsts adam, int *nonfinite_step, int step)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= v.n - 1) return;
    const int lo = v.lo[k], mid = v.mid[k], hi1 = v.hi1[k];
    const double Hr = dd_diff(C[mid], C[lo]), Hl = dd_diff(C[hi1], C[mid]);
    const double y = ys[k];
    const double zg = y * (1 - y) * (Hl / y - Hr / (1.0 - y));
    if (!isfinite(zg)) atomicCAS(nonfinite_step, 0, step);
    float m = m_z[k], vv = v_z[k], p = zs[k];
    if (adam.first) {
        m = (float)zg;
        vv = (float)(zg * zg);
    } else {
        m = (float)(adam.rm * (double)m + (1 - adam.rm) * zg);
        vv = (float)(adam.rv * (double)vv + (1 - adam.rv) * (zg * zg));
    }
    const double pm = (double)m / adam.m_denom, pv = (double)vv / adam.v_denom;
    double delta = adam.lr * pm / (sqrt(pv) + adam.eps);
    delta = delta < -adam.max_mu ? -adam.max_mu : (delta > adam.max_mu ? adam.max_mu : delta);
    zs[k] = (float)((double)p + delta);
    m_z[k] = m;
    v_z[k] = vv;
}

// sum of the per-chunk log-u partials -> row_sums[d*2+1] (hsb ladj), and csum -> row_sums[d*2+0]
template <int K>
__global__ void vi_values_finish_kernel(const double *part_ladj, int nchunks, const double *csum, double *row_sums)
{
    __shared__ double smd[4 * K];
    double s[K];
#pragma unroll
    for (int d = 0; d < K; ++d) s[d] = 0.0;
    for (int ch = threadIdx.x; ch < nchunks; ch += blockDim.x)
#pragma unroll
        for (int d = 0; d < K; ++d) s[d] += part_ladj[(size_t)ch * K + d];
    block_sum_vec<K>(s, smd);
#pragma unroll
    for (int d = 0; d < K; ++d)
        if (threadIdx.x == d) {
            row_sums[d * 2 + 1] = s[d];
            row_sums[d * 2 + 0] = csum ? csum[d] : 0.0;
        }
}

}  // namespace polee
