// Construction of the likelihood matrix X on the GPU (first slice): from alignment pairs
// (intervals + CIGAR operations) and transcripts (exon intervals) to the compressed rows of X -- which fragments are
// compatible with which transcripts, with the conditional fragment probability of the reference's SimplisticFragModel
// (bias terms = 1) -- ready for polee_loglik_create_from_xt.
//
// Replaces, for pre-parsed inputs:
//   parallel_intersection_loop          src/rnaseq_sample.jl:58-121   (interval-tree join + condfragprob per pair)
//   fragmentlength                      src/transcripts.jl:273-446    (CIGAR intervals against exons / introns)
//   effective_length, condfragprob      src/fragmodel.jl:119-169      (SimplisticFragModel)
//   sortperm / compact_indexes! / sparse  src/rnaseq_sample.jl:126-157, 470-489
// BAM / GFF parsing, bias models and read assignment stay upstream (out of scope).
//
// Formulation: the reference joins two interval trees per sequence on threads.  Here a thread owns a FRAGMENT: the
// transcripts of its sequence are sorted by first base (once, on the host: a few hundred thousand of them) with a running
// maximum of their last bases, so the transcripts containing the fragment are a backward scan from a binary search that
// stops as soon as no earlier transcript can reach the fragment's end.  Two passes (count, then fill behind an exclusive
// prefix sum of the counts) produce the rows without atomics; rows are emitted in fragment order, ids ascending inside
// a row, fragments without any entry dropped (compact_indexes!).  Everything is integer interval logic plus one table
// lookup and a few float operations per pair: latency-bound gathers, no roofline claim.
// Precision: as the reference -- Float32 products for a fragment on the transcript's strand, Float64 when it is not
// (`1.0 - strand_specificity`), the effective length accumulated in Float32 in sequence (one thread per transcript, so
// that the sum is the same sum); see oracle/xbuild_oracle.c for the quirks kept and not kept.
#include "common.hpp"
#include "scan.hpp"

#include <algorithm>
#include <numeric>

// The reference's Float32 / Float64 products and sums are separate roundings (Julia does not contract a * b + c into an
// fma).  hipcc's default is -ffp-contract=fast, under which the backend fuses even __fmul_rn / __fadd_rn pairs (1-ulp
// differences in a third of the effective lengths): this file is compiled with -ffp-contract=off (csrc/Makefile).
#pragma clang fp contract(off)

namespace polee {

constexpr int XB_MAX_FRAG_LEN = 2000;            // src/constants.jl:34
constexpr float XB_MIN_EFFECTIVE_LENGTH = 1.0f;  // src/constants.jl:41
constexpr double XB_MIN_FRAG_PROB = 1e-12;       // src/constants.jl:45
enum : int { XB_MATCH = 0, XB_INSERT = 1, XB_DELETE = 2, XB_SKIP = 3, XB_SOFT_CLIP = 4 };  // BAM operation codes

struct XbView {
    // transcripts
    int32_t n;
    const int32_t *t_seq;
    const int8_t *t_strand;
    const int64_t *exon_ptr, *exon_first, *exon_last;
    // transcripts of every sequence sorted by first base: order [n] -> transcript, first [n], running max of last [n]
    const int32_t *ord;
    const int64_t *ord_first, *ord_maxlast;
    const int64_t *seq_ptr;  // [num_seq + 1] into ord
    int32_t num_seq;
    // fragments
    int64_t m;
    const int32_t *f_seq;
    const int8_t *f_strand;
    const int64_t *m1_left, *m1_right, *m2_left, *m2_right;
    const uint8_t *m1_is_flag16;
    const int64_t *cig1_ptr, *cig2_ptr;
    const uint8_t *cig_op;
    const int32_t *cig_len;
    // fragment model
    const float *pmf, *cdf;
    int32_t fraglen_median;
    float strand_specificity;
    int32_t alt_frag_model;
    const float *efflens;
};

struct CigIter {
    const uint8_t *op;
    const int32_t *len;
    int64_t cnt, i, pos, left, right;
};
struct CigIv {
    int64_t first, last;
    int op;
};
// CigarIter (src/reads.jl:458-492)
__device__ inline bool cig_next(CigIter &it, CigIv &out)
{
    if (it.cnt == 0) {
        if (it.i > 0) return false;
        it.i = 1;
        out.first = it.left;
        out.last = it.right;
        out.op = XB_MATCH;
        return true;
    }
    if (it.i >= it.cnt) return false;
    out.op = it.op[it.i];
    out.first = it.pos;
    out.last = it.pos + it.len[it.i] - 1;
    it.pos += it.len[it.i];
    ++it.i;
    return true;
}
__device__ inline bool exon_compatible(int op) { return op == XB_MATCH || op == XB_SOFT_CLIP || op == XB_INSERT || op == XB_DELETE; }  // reads.jl:510
__device__ inline bool intron_compatible(int op) { return op == XB_SKIP || op == XB_SOFT_CLIP; }                                      // reads.jl:516
// reads.jl:521-537
__device__ inline void next_exonintron(const int64_t *ef, const int64_t *el, int64_t ne, int64_t &idx, bool &is_exon, int64_t &first, int64_t &last)
{
    if (is_exon) {
        if (idx + 1 < ne) {
            first = el[idx] + 1;
            last = ef[idx + 1] - 1;
        } else {
            idx += 1;
        }
    } else {
        idx += 1;
        first = ef[idx];
        last = el[idx];
    }
    is_exon = !is_exon;
}

// one alignment's CIGAR intervals against the exons / introns from `first_idx` on: false = incompatible.
// SECOND: the rightmost mate (introns passed AFTER the exon / intron where the first mate's walk ended count as spanned)
template <bool SECOND>
__device__ inline bool walk_mate(const int64_t *ef, const int64_t *el, int64_t ne, int64_t first_idx, CigIter &ci, int64_t &intronlen,
                                 int64_t &e_idx, int64_t &e_first, int64_t &e_last, int64_t e1_idx, int64_t e1_first, int64_t e1_last)
{
    constexpr int64_t max_enc = 2;  // matches overhanging into an intron by <= 2 bases are allowed (transcripts.jl:275)
    CigIv c;
    bool have = cig_next(ci, c);
    e_idx = first_idx;
    e_first = ef[e_idx];
    e_last = el[e_idx];
    bool e_isexon = true, sup = false;
    if (!SECOND && have && c.op == XB_SOFT_CLIP) have = cig_next(ci, c);  // leading soft clipping (:312)
    while (e_idx < ne && have) {
        if (e_last < c.first) {  // case 1: the exon / intron entirely precedes
            if (SECOND) {
                if (!e_isexon && sup) intronlen += e_last - e_first + 1;
                if (e1_idx < ne && e1_first == e_first && e1_last == e_last) sup = true;
            }
            next_exonintron(ef, el, ne, e_idx, e_isexon, e_first, e_last);
        } else if (c.last >= e_first && c.last <= e_last && c.first >= e_first) {  // case 2: contained
            if (e_isexon) {
                if (!exon_compatible(c.op)) return false;
            } else {
                if (!intron_compatible(c.op)) return false;
                if (!SECOND) intronlen += e_last - e_first + 1;
            }
            have = cig_next(ci, c);
        } else if (c.op == XB_SOFT_CLIP) {  // case 3
            have = cig_next(ci, c);
        } else if (c.last > e_last && c.op == XB_MATCH) {  // case 4: a match overhangs a little
            if (e_isexon && c.last - e_last <= max_enc) c.last = e_last;
            else if (!e_isexon && e_last >= c.first && e_last - c.first < max_enc) c.first = e_last + 1;
            else return false;
        } else {
            return false;  // case 5
        }
    }
    if (SECOND && have && c.op == XB_SOFT_CLIP) have = cig_next(ci, c);  // trailing soft clipping (:430)
    return !have;
}

// fragmentlength (src/transcripts.jl:273-446): -1 incompatible, 0 compatible single-end, > 0 the fragment's length
__device__ inline int64_t fragmentlength(const XbView &v, int32_t j, int64_t i)
{
    const int64_t e0 = v.exon_ptr[j], ne = v.exon_ptr[j + 1] - e0;
    const int64_t *ef = v.exon_first + e0, *el = v.exon_last + e0;
    const bool paired = v.m2_left[i] > 0;
    const int64_t a_first = v.m1_left[i];
    int64_t a_last = v.m1_right[i];
    if (paired && v.m2_right[i] > a_last) a_last = v.m2_right[i];
    if (a_first < ef[0] || a_last > el[ne - 1]) return -1;
    // searchsortedlast(exons, alnpr); at least the first exon (oracle/xbuild_oracle.c, "quirk not kept")
    int64_t first_idx = 0;
    {
        int64_t lo = 0, hi = ne;  // first exon k with (ef, el)[k] > (a_first, a_last)
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ef[mid] < a_first || (ef[mid] == a_first && el[mid] <= a_last)) lo = mid + 1; else hi = mid;
        }
        first_idx = lo > 0 ? lo - 1 : 0;
    }
    int64_t intronlen = 0, e1_idx, e1_first, e1_last, e2_idx, e2_first, e2_last;
    CigIter c1{v.cig_op + v.cig1_ptr[i], v.cig_len + v.cig1_ptr[i], v.cig1_ptr[i + 1] - v.cig1_ptr[i], 0, v.m1_left[i], v.m1_left[i], v.m1_right[i]};
    if (!walk_mate<false>(ef, el, ne, first_idx, c1, intronlen, e1_idx, e1_first, e1_last, 0, 0, 0)) return -1;
    if (!paired) return 0;
    CigIter c2{v.cig_op + v.cig2_ptr[i], v.cig_len + v.cig2_ptr[i], v.cig2_ptr[i + 1] - v.cig2_ptr[i], 0, v.m2_left[i], v.m2_left[i], v.m2_right[i]};
    if (!walk_mate<true>(ef, el, ne, first_idx, c2, intronlen, e2_idx, e2_first, e2_last, e1_idx, e1_first, e1_last)) return -1;
    const int64_t rmax = max(v.m1_right[i], v.m2_right[i]), lmin = min(v.m1_left[i], v.m2_left[i]);
    const int64_t fraglen = rmax - lmin + 1 - intronlen;
    return fraglen > 0 ? fraglen : -1;
}

__device__ inline int64_t exonic_length(const XbView &v, int32_t j)
{
    int64_t s = 0;
    for (int64_t k = v.exon_ptr[j]; k < v.exon_ptr[j + 1]; ++k) s += v.exon_last[k] - v.exon_first[k] + 1;
    return s;
}

// effective_length(::SimplisticFragModel, t) (src/fragmodel.jl:155-169): a Float32 sum in sequence, as the reference's
__global__ void xb_efflen_kernel(XbView v, float *__restrict__ efflens)
{
    const int32_t j = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= v.n) return;
    const int64_t tlen = exonic_length(v, j);
    const int64_t top = min(tlen, (int64_t)XB_MAX_FRAG_LEN);
    float r;
    if (v.alt_frag_model && tlen > XB_MAX_FRAG_LEN) {  // denom = 1.0 (Float64)
        double el = 0.0;
        for (int64_t l = 1; l <= top; ++l) el = __dadd_rn(el, __dmul_rn((double)v.pmf[l - 1], (double)(tlen - l + 1)));
        r = (float)fmax(el, (double)XB_MIN_EFFECTIVE_LENGTH);
    } else {
        float el = 0.0f;
        if (v.alt_frag_model) {
            const float denom = v.cdf[tlen - 1];
            for (int64_t l = 1; l <= top; ++l) el = __fadd_rn(el, __fmul_rn(__fdiv_rn(v.pmf[l - 1], denom), (float)(tlen - l + 1)));
        } else {
            for (int64_t l = 1; l <= top; ++l) el = __fadd_rn(el, __fmul_rn(v.pmf[l - 1], (float)(tlen - l + 1)));
        }
        r = fmaxf(el, XB_MIN_EFFECTIVE_LENGTH);
    }
    efflens[j] = r;
}

// condfragprob(::SimplisticFragModel, ...) (src/fragmodel.jl:119-153)
__device__ inline float condfragprob(const XbView &v, int32_t j, int64_t i)
{
    int64_t fraglen = fragmentlength(v, j, i);
    if (fraglen < 0) return 0.0f;
    const int64_t e0 = v.exon_ptr[j], e1 = v.exon_ptr[j + 1];
    if (fraglen <= 0) {  // single-end read
        const int64_t maxlen = v.m1_is_flag16[i] ? v.m1_right[i] - v.exon_first[e0] + 1 : v.exon_last[e1 - 1] - v.m1_left[i] + 1;
        fraglen = min(maxlen, (int64_t)v.fraglen_median);
    }
    const float fraglenpr = fraglen >= 1 && fraglen <= XB_MAX_FRAG_LEN ? v.pmf[fraglen - 1] : 0.0f;
    const float efflen = v.efflens[j];
    const bool same = v.f_strand[i] == v.t_strand[j];
    double fragpr;
    if (same) fragpr = (double)__fdiv_rn(__fmul_rn(v.strand_specificity, fraglenpr), efflen);
    else fragpr = __ddiv_rn(__dmul_rn(1.0 - (double)v.strand_specificity, (double)fraglenpr), (double)efflen);
    if (v.alt_frag_model) {
        const int64_t tlen = exonic_length(v, j);
        if (tlen <= XB_MAX_FRAG_LEN) {
            if (same) fragpr = (double)__fdiv_rn((float)fragpr, v.cdf[tlen - 1]);
            else fragpr = __ddiv_rn(fragpr, (double)v.cdf[tlen - 1]);
        }
    }
    return (float)fragpr;
}

// Transcripts containing fragment i, in descending order of their position in the sorted list; f(j, fragpr) for the kept
// ones (finite, > MIN_FRAG_PROB: rnaseq_sample.jl:99)
template <typename F>
__device__ inline void for_each_entry(const XbView &v, int64_t i, F &&f)
{
    const int32_t s = v.f_seq[i];
    if (s < 0 || s >= v.num_seq) return;
    const int64_t b = v.seq_ptr[s], e = v.seq_ptr[s + 1];
    const int64_t a_first = v.m1_left[i];
    int64_t a_last = v.m1_right[i];
    if (v.m2_left[i] > 0 && v.m2_right[i] > a_last) a_last = v.m2_right[i];
    int64_t lo = b, hi = e;  // first position with first base > a_first
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (v.ord_first[mid] <= a_first) lo = mid + 1; else hi = mid;
    }
    for (int64_t k = lo - 1; k >= b; --k) {
        if (v.ord_maxlast[k] < a_last) break;  // no transcript at or before k reaches the fragment's end
        const int32_t j = v.ord[k];
        if (v.exon_last[v.exon_ptr[j + 1] - 1] < a_last) continue;  // intersect_contains (rnaseq_sample.jl:77-79)
        const float p = condfragprob(v, j, i);
        if (isfinite(p) && (double)p > XB_MIN_FRAG_PROB) f(j, p);
    }
}

__global__ void xb_count_kernel(XbView v, int64_t *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.m) return;
    int64_t c = 0;
    for_each_entry(v, i, [&](int32_t, float) { ++c; });
    counts[i] = c;
}

// rows: entries at off[i] .. off[i+1], then sorted by transcript id (rows are short)
__global__ void xb_fill_kernel(XbView v, const int64_t *__restrict__ off, uint32_t *__restrict__ cols, float *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.m) return;
    const int64_t b = off[i], e = off[i + 1];
    if (e == b) return;
    int64_t w = b;
    for_each_entry(v, i, [&](int32_t j, float p) {
        if (w < e) {
            cols[w] = (uint32_t)j + 1u;
            vals[w] = p;
            ++w;
        }
    });
    for (int64_t p = b + 1; p < e; ++p) {  // insertion sort
        const uint32_t c = cols[p];
        const float x = vals[p];
        int64_t q = p;
        while (q > b && cols[q - 1] > c) {
            cols[q] = cols[q - 1];
            vals[q] = vals[q - 1];
            --q;
        }
        cols[q] = c;
        vals[q] = x;
    }
}

// ---- exclusive prefix sums of int64 (three launches: chunk totals, their scan by one workgroup, apply) ----------------
constexpr int XS_T = 256, XS_ITEMS = 8, XS_CHUNK = XS_T * XS_ITEMS;
__device__ inline int64_t block_excl_scan_i64(int64_t v, int64_t *smem, int64_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    int64_t base = 0, tot = 0;
    for (int w = 0; w < XS_T / 64; ++w) {
        if (w < wave) base += smem[w];
        tot += smem[w];
    }
    __syncthreads();
    if (total) *total = tot;
    return base + incl - v;
}
template <bool FLAGS>  // FLAGS: scan (in[i] > 0) instead of in[i]
__global__ __launch_bounds__(XS_T) void xs_reduce_kernel(const int64_t *__restrict__ in, int64_t n, int64_t *__restrict__ chunk_sums)
{
    __shared__ int64_t smem[XS_T / 64];
    const int64_t base = (int64_t)blockIdx.x * XS_CHUNK + (int64_t)threadIdx.x * XS_ITEMS;
    int64_t acc = 0;
    for (int k = 0; k < XS_ITEMS; ++k)
        if (base + k < n) acc += FLAGS ? (in[base + k] > 0 ? 1 : 0) : in[base + k];
    int64_t tot;
    (void)block_excl_scan_i64(acc, smem, &tot);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(XS_T) void xs_spine_kernel(int64_t *chunk_sums, int64_t nchunks, int64_t *grand_total)
{
    __shared__ int64_t smem[XS_T / 64];
    int64_t carry = 0;
    for (int64_t b = 0; b < nchunks; b += XS_T) {
        const int64_t i = b + threadIdx.x;
        const int64_t v = i < nchunks ? chunk_sums[i] : 0;
        int64_t tot;
        const int64_t ex = block_excl_scan_i64(v, smem, &tot);
        if (i < nchunks) chunk_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
template <bool FLAGS>
__global__ __launch_bounds__(XS_T) void xs_apply_kernel(const int64_t *__restrict__ in, int64_t n, const int64_t *__restrict__ chunk_off, int64_t *__restrict__ out)
{
    __shared__ int64_t smem[XS_T / 64];
    const int64_t base = (int64_t)blockIdx.x * XS_CHUNK + (int64_t)threadIdx.x * XS_ITEMS;
    int64_t vals[XS_ITEMS], acc = 0;
    for (int k = 0; k < XS_ITEMS; ++k) {
        vals[k] = base + k < n ? (FLAGS ? (in[base + k] > 0 ? 1 : 0) : in[base + k]) : 0;
        acc += vals[k];
    }
    int64_t run = chunk_off[blockIdx.x] + block_excl_scan_i64(acc, smem, nullptr);
    for (int k = 0; k < XS_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += vals[k];
    }
    if (base <= n && n < base + XS_ITEMS) out[n] = run - 0;  // (the thread whose range holds index n writes the total)
}
template <bool FLAGS>
static polee_status exclusive_scan_i64(polee_ctx *ctx, const int64_t *d_in, int64_t n, int64_t *d_out /* [n+1] */, DevBuf<int64_t> &tmp, int64_t *h_total)
{
    const int64_t nchunks = std::max<int64_t>(1, (n + 1 + XS_CHUNK - 1) / XS_CHUNK);
    POLEE_TRY(tmp.alloc(ctx, (size_t)nchunks + 1));
    hipLaunchKernelGGL((xs_reduce_kernel<FLAGS>), dim3((unsigned)nchunks), dim3(XS_T), 0, ctx->stream, d_in, n, tmp.p);
    hipLaunchKernelGGL(xs_spine_kernel, dim3(1), dim3(XS_T), 0, ctx->stream, tmp.p, nchunks, tmp.p + nchunks);
    hipLaunchKernelGGL((xs_apply_kernel<FLAGS>), dim3((unsigned)nchunks), dim3(XS_T), 0, ctx->stream, d_in, n, tmp.p, d_out);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_HIP_TRY(ctx, hipMemcpyAsync(h_total, tmp.p + nchunks, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return POLEE_OK;
}

// kept rows: tcolptr (1-based) and the fragment of every row
__global__ void xb_rows_kernel(const int64_t *__restrict__ off, const int64_t *__restrict__ rowid, int64_t m, uint64_t *__restrict__ tcolptr,
                               int64_t *__restrict__ row_fragment)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) tcolptr[0] = 1;
    if (i >= m) return;
    if (off[i + 1] > off[i]) {
        const int64_t r = rowid[i];
        tcolptr[r + 1] = (uint64_t)off[i + 1] + 1u;
        row_fragment[r] = i;
    }
}

}  // namespace polee

using namespace polee;

struct polee_xbuild {
    polee_ctx *ctx = nullptr;
    int64_t rows = 0, nnz = 0, m = 0;
    int32_t n = 0;
    DevBuf<float> d_efflens, d_vals;
    DevBuf<uint32_t> d_cols;
    DevBuf<uint64_t> d_tcolptr;
    DevBuf<int64_t> d_row_fragment;
    double ms_efflen = 0, ms_count = 0, ms_fill = 0;
};

extern "C" {

static polee_status polee_xbuild_run_impl(polee_ctx *ctx, const polee_xb_transcripts *T, const polee_xb_fragments *F, const polee_xb_fragmodel *M,
                              polee_xbuild **out)
{
    POLEE_TRY(use_device(ctx));
    if (!T || !F || !M || !out) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_xbuild_run: null argument");
    if (T->n < 1 || F->m < 0 || !T->seq || !T->strand || !T->exon_ptr || !T->exon_first || !T->exon_last || !M->fraglen_pmf || !M->fraglen_cdf)
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_xbuild_run: bad argument");
    const int32_t n = T->n;
    const int64_t m = F->m, nex = T->exon_ptr[n];
    // fragments: every array present, mates ordered as the reference orders them itself (a1 = the leftmost mate,
    // transcripts.jl:288-297 -- a caller holding them in BAM order swaps first, polee_amd/xbuild.py does), intervals and
    // CIGAR ranges well formed -- the kernels index with them
    if (m > 0 && (!F->seq || !F->strand || !F->m1_left || !F->m1_right || !F->m2_left || !F->m2_right || !F->m1_is_flag16 || !F->cig1_ptr))
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_xbuild_run: a fragment array is null (only cig2_ptr may be: all single-end)");
    std::vector<int64_t> cig2_zero;  // cig2_ptr == NULL: no second mate has CIGAR operations
    const int64_t *cig2 = F->cig2_ptr;
    if (m > 0 && !cig2) {
        cig2_zero.assign((size_t)m + 1, 0);
        cig2 = cig2_zero.data();
    }
    const int64_t ncig = m > 0 ? std::max(F->cig1_ptr[m], cig2[m]) : 0;
    if (m > 0 && (F->cig1_ptr[0] < 0 || cig2[0] < 0 || (ncig > 0 && (!F->cig_op || !F->cig_len))))
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_xbuild_run: bad CIGAR offsets");
    {
        std::atomic<int> bad{0};
        std::atomic<int64_t> where{-1};
        parallel_chunks((size_t)m, (size_t)1 << 18, [&](size_t lo, size_t hi, unsigned) {
            for (size_t i = lo; i < hi; ++i) {
                int why = 0;
                if (F->seq[i] < 0) why = 1;
                else if (F->m1_left[i] < 1 || F->m1_right[i] < F->m1_left[i]) why = 2;
                else if (F->m2_left[i] != 0 && (F->m2_left[i] < 1 || F->m2_right[i] < F->m2_left[i])) why = 2;
                else if (F->m2_left[i] != 0 && F->m2_left[i] < F->m1_left[i]) why = 3;
                else if (F->cig1_ptr[i + 1] < F->cig1_ptr[i] || cig2[i + 1] < cig2[i]) why = 4;
                else if (F->m2_left[i] == 0 && cig2[i + 1] != cig2[i]) why = 4;
                else {
                    for (int64_t k = F->cig1_ptr[i]; k < F->cig1_ptr[i + 1] && !why; ++k)
                        if (F->cig_len[k] < 0 || F->cig_op[k] > 8) why = 5;
                    for (int64_t k = cig2[i]; k < cig2[i + 1] && !why; ++k)
                        if (F->cig_len[k] < 0 || F->cig_op[k] > 8) why = 5;
                }
                if (why) {
                    int z = 0;
                    if (bad.compare_exchange_strong(z, why)) where = (int64_t)i;
                    return;
                }
            }
        });
        static const char *const msg[] = {"", "negative sequence id", "mate interval is not 1-based with left <= right",
                                          "m1 must be the LEFTMOST mate (the reference orders the mates by leftpos, transcripts.jl:288-297)",
                                          "CIGAR offsets are not monotone (or a single-end fragment has operations for a second mate)",
                                          "CIGAR operation out of range or negative length"};
        if (bad) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_xbuild_run: fragment %lld: %s", (long long)where.load(), msg[bad.load()]);
    }
    // transcripts: every one needs exons, ascending and disjoint
    int32_t num_seq = 0;
    for (int32_t j = 0; j < n; ++j) {
        if (T->exon_ptr[j + 1] <= T->exon_ptr[j]) return fail(ctx, POLEE_ERR_BAD_ARG, "transcript %d has no exons", j);
        for (int64_t k = T->exon_ptr[j]; k < T->exon_ptr[j + 1]; ++k)
            if (T->exon_last[k] < T->exon_first[k] || (k > T->exon_ptr[j] && T->exon_first[k] <= T->exon_last[k - 1]))
                return fail(ctx, POLEE_ERR_BAD_ARG, "exons of transcript %d are not ascending and disjoint", j);
        if (T->seq[j] < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "negative sequence id");
        num_seq = std::max(num_seq, T->seq[j] + 1);
    }
    // per sequence: transcripts sorted by first base, running maximum of the last base
    std::vector<int32_t> ord((size_t)n);
    std::iota(ord.begin(), ord.end(), 0);
    auto tfirst = [&](int32_t j) { return T->exon_first[T->exon_ptr[j]]; };
    auto tlast = [&](int32_t j) { return T->exon_last[T->exon_ptr[j + 1] - 1]; };
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
        return T->seq[a] != T->seq[b] ? T->seq[a] < T->seq[b] : tfirst(a) < tfirst(b);
    });
    std::vector<int64_t> ofirst((size_t)n), omax((size_t)n), seq_ptr((size_t)num_seq + 1, 0);
    for (int32_t k = 0; k < n; ++k) {
        const int32_t j = ord[(size_t)k];
        ofirst[(size_t)k] = tfirst(j);
        const bool fresh = k == 0 || T->seq[ord[(size_t)k - 1]] != T->seq[j];
        omax[(size_t)k] = fresh ? tlast(j) : std::max(omax[(size_t)k - 1], tlast(j));
        ++seq_ptr[(size_t)T->seq[j] + 1];
    }
    for (int32_t s = 0; s < num_seq; ++s) seq_ptr[(size_t)s + 1] += seq_ptr[(size_t)s];

    polee_xbuild *xb = new (std::nothrow) polee_xbuild();
    if (!xb) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    xb->ctx = ctx;
    xb->n = n;
    xb->m = m;
    ctx_retain(ctx);
    DevBuf<int32_t> d_tseq, d_ord, d_fseq, d_ciglen;
    DevBuf<int8_t> d_tstrand, d_fstrand;
    DevBuf<int64_t> d_eptr, d_ef, d_el, d_ofirst, d_omax, d_seqptr, d_m1l, d_m1r, d_m2l, d_m2r, d_c1, d_c2, d_counts, d_off, d_rowid, d_tmp;
    DevBuf<uint8_t> d_flag16, d_cigop;
    DevBuf<float> d_pmf, d_cdf;
    polee_status s = POLEE_OK;
    auto A = [&](polee_status r) {
        if (s == POLEE_OK) s = r;
    };
    A(d_tseq.upload(ctx, T->seq, (size_t)n)); A(d_tstrand.upload(ctx, T->strand, (size_t)n));
    A(d_eptr.upload(ctx, T->exon_ptr, (size_t)n + 1)); A(d_ef.upload(ctx, T->exon_first, (size_t)nex)); A(d_el.upload(ctx, T->exon_last, (size_t)nex));
    A(d_ord.upload(ctx, ord)); A(d_ofirst.upload(ctx, ofirst)); A(d_omax.upload(ctx, omax)); A(d_seqptr.upload(ctx, seq_ptr));
    A(d_fseq.upload(ctx, F->seq, (size_t)m)); A(d_fstrand.upload(ctx, F->strand, (size_t)m));
    A(d_m1l.upload(ctx, F->m1_left, (size_t)m)); A(d_m1r.upload(ctx, F->m1_right, (size_t)m));
    A(d_m2l.upload(ctx, F->m2_left, (size_t)m)); A(d_m2r.upload(ctx, F->m2_right, (size_t)m));
    A(d_flag16.upload(ctx, F->m1_is_flag16, (size_t)m));
    A(d_c1.upload(ctx, F->cig1_ptr, (size_t)m + 1)); A(d_c2.upload(ctx, cig2, (size_t)m + 1));
    A(d_cigop.upload(ctx, F->cig_op, (size_t)ncig)); A(d_ciglen.upload(ctx, F->cig_len, (size_t)ncig));
    A(d_pmf.upload(ctx, M->fraglen_pmf, (size_t)XB_MAX_FRAG_LEN)); A(d_cdf.upload(ctx, M->fraglen_cdf, (size_t)XB_MAX_FRAG_LEN));
    A(xb->d_efflens.alloc(ctx, (size_t)n)); A(d_counts.alloc(ctx, (size_t)m + 1)); A(d_off.alloc(ctx, (size_t)m + 2)); A(d_rowid.alloc(ctx, (size_t)m + 2));
    if (s != POLEE_OK) {
        polee_xbuild_destroy(xb);
        return s;
    }
    XbView v{n, d_tseq.p, d_tstrand.p, d_eptr.p, d_ef.p, d_el.p, d_ord.p, d_ofirst.p, d_omax.p, d_seqptr.p, num_seq, m, d_fseq.p, d_fstrand.p,
             d_m1l.p, d_m1r.p, d_m2l.p, d_m2r.p, d_flag16.p, d_c1.p, d_c2.p, d_cigop.p, d_ciglen.p, d_pmf.p, d_cdf.p, M->fraglen_median,
             M->strand_specificity, M->alt_frag_model, xb->d_efflens.p};
    hipStream_t st = ctx->stream;
    auto timed = [&](double &ms, auto &&launch) -> polee_status {
        (void)hipEventRecord(ctx->ev0, st);
        launch();
        (void)hipEventRecord(ctx->ev1, st);
        POLEE_HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float t = 0.f;
        POLEE_HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
        ms = t;
        POLEE_KERNEL_CHECK(ctx);
        return POLEE_OK;
    };
    const unsigned nbm = (unsigned)std::max<int64_t>(1, ceil_div(m, 256));
    int64_t nnz = 0, rows = 0;
    if ((s = timed(xb->ms_efflen, [&] { hipLaunchKernelGGL(xb_efflen_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, st, v, xb->d_efflens.p); })) ||
        (s = timed(xb->ms_count, [&] { hipLaunchKernelGGL(xb_count_kernel, dim3(nbm), dim3(256), 0, st, v, d_counts.p); })) ||
        (s = exclusive_scan_i64<false>(ctx, d_counts.p, m, d_off.p, d_tmp, &nnz)) ||
        (s = exclusive_scan_i64<true>(ctx, d_counts.p, m, d_rowid.p, d_tmp, &rows)) ||
        (s = xb->d_cols.alloc(ctx, (size_t)std::max<int64_t>(nnz, 1))) || (s = xb->d_vals.alloc(ctx, (size_t)std::max<int64_t>(nnz, 1))) ||
        (s = xb->d_tcolptr.alloc(ctx, (size_t)rows + 1)) || (s = xb->d_row_fragment.alloc(ctx, (size_t)std::max<int64_t>(rows, 1))) ||
        (s = timed(xb->ms_fill, [&] {
            hipLaunchKernelGGL(xb_fill_kernel, dim3(nbm), dim3(256), 0, st, v, d_off.p, xb->d_cols.p, xb->d_vals.p);
            hipLaunchKernelGGL(xb_rows_kernel, dim3(nbm), dim3(256), 0, st, d_off.p, d_rowid.p, m, xb->d_tcolptr.p, xb->d_row_fragment.p);
        }))) {
        polee_xbuild_destroy(xb);
        return s;
    }
    xb->nnz = nnz;
    xb->rows = rows;
    *out = xb;
    return POLEE_OK;
}

polee_status polee_xbuild_run(polee_ctx *ctx, const polee_xb_transcripts *T, const polee_xb_fragments *F, const polee_xb_fragmodel *M,
                              polee_xbuild **out)
{
    return guarded(ctx, "polee_xbuild_run", [&] { return polee_xbuild_run_impl(ctx, T, F, M, out); });
}

void polee_xbuild_destroy(polee_xbuild *xb)
{
    if (!xb) return;
    polee_ctx *ctx = xb->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    delete xb;
    ctx_release(ctx);
}

polee_status polee_xbuild_sizes(const polee_xbuild *xb, int64_t *rows, int64_t *nnz, double *ms_efflen, double *ms_count, double *ms_fill)
{
    if (!xb) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    if (rows) *rows = xb->rows;
    if (nnz) *nnz = xb->nnz;
    if (ms_efflen) *ms_efflen = xb->ms_efflen;
    if (ms_count) *ms_count = xb->ms_count;
    if (ms_fill) *ms_fill = xb->ms_fill;
    return POLEE_OK;
}

polee_status polee_xbuild_get(const polee_xbuild *xb, uint64_t *tcolptr, uint32_t *trowval, float *tnzval, float *effective_lengths,
                              int64_t *row_fragment)
{
    if (!xb) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = xb->ctx;
    POLEE_TRY(use_device(ctx));
    if (tcolptr) POLEE_TRY(xb->d_tcolptr.download(ctx, tcolptr, (size_t)xb->rows + 1));
    if (trowval) POLEE_TRY(xb->d_cols.download(ctx, trowval, (size_t)xb->nnz));
    if (tnzval) POLEE_TRY(xb->d_vals.download(ctx, tnzval, (size_t)xb->nnz));
    if (effective_lengths) POLEE_TRY(xb->d_efflens.download(ctx, effective_lengths, (size_t)xb->n));
    if (row_fragment) POLEE_TRY(xb->d_row_fragment.download(ctx, row_fragment, (size_t)xb->rows));
    return POLEE_OK;
}

}  // extern "C"
