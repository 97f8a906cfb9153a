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
// Round 4: the same with the reference's DEFAULT BiasedFragModel given a TRAINED bias model (polee_xbuild_run_biased): the
// transcripts' bias vectors (compute_transcript_bias!, bias.jl:834-858), the biased effective lengths (fragmodel.jl:372-410)
// and condfragprob with the fragment's interval on the transcript (fragmodel.jl:413-445, transcripts.jl:452-538).
// BAM / GFF parsing, TRAINING of the bias model and read assignment stay upstream (out of scope).
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
    // BiasedFragModel (null tseq: the SimplisticFragModel): sequences, the trained bias model's tables, the transcripts'
    // bias vectors (compute_transcript_bias!, filled by xb_bias_kernel)
    const int64_t *tseq_ptr;
    const uint8_t *tseq;
    int32_t seqbias_len, ps_ctx;
    const int32_t *orders_left, *orders_right;
    const float *ps_left, *ps_right;
    int32_t gc_nbins;
    const float *gc_bins;
    double pos_p;
    const double *pos_terms;
    int32_t num_fraglens;
    const int32_t *high_prob_fraglens;
    const uint8_t *m1_reverse;
    float *left_bias, *right_bias;
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

// ---- BiasedFragModel (src/fragmodel.jl:174-445 with a TRAINED bias model, src/bias.jl:402-456, 517-520, 649-663, 834-858) ---
constexpr int XB_BIAS_SEQ_INNER_CTX = 15, XB_BIAS_SEQ_OUTER_CTX = 5;  // src/constants.jl:77-78
__device__ inline int xb_code_at(const uint8_t *seq, int64_t len, int64_t j)  // 1-based; off the ends: A (the reference draws a random
{                                                                             // nucleotide there, bias.jl:424-429: not reproducible)
    if (j < 1 || j > len) return 0;
    const uint8_t c = seq[j - 1];
    return c < 4 ? c : 0;  // nt2bit: N -> 0 (bias.jl:176-181)
}
__device__ inline int xb_is_gc(uint8_t c) { return c == 1 || c == 2; }
template <bool RIGHT>
__device__ inline float xb_seqbias_eval(const XbView &v, const uint8_t *seq, int64_t len, int64_t pos)
{
    const int32_t *orders = RIGHT ? v.orders_right : v.orders_left;
    const float *ps = RIGHT ? v.ps_right : v.ps_left;
    const int64_t first = RIGHT ? pos - XB_BIAS_SEQ_INNER_CTX + 1 : pos - XB_BIAS_SEQ_OUTER_CTX;
    float bias = 1.0f;
    for (int32_t i = 0; i < v.seqbias_len; ++i) {
        const int32_t order = orders[i];
        if (order < 0) continue;
        const int64_t j = first + i;
        const int c = xb_code_at(seq, len, j);
        int ctx = 0;
        for (int l = 1; l <= order; ++l) ctx = (ctx << 2) | xb_code_at(seq, len, j + l);
        bias = __fmul_rn(bias, ps[((size_t)i * 4 + (size_t)c) * (size_t)v.ps_ctx + (size_t)ctx]);
    }
    return bias;
}
__device__ inline float xb_hist_f32(const XbView &v, float x)  // evaluate(::SimpleHistogramModel, x): round(Int, x * nbins) to even
{
    long long i = __float2ll_rn(__fmul_rn(x, (float)v.gc_nbins));
    i = i < 1 ? 1 : (i > v.gc_nbins ? v.gc_nbins : i);
    return v.gc_bins[i - 1];
}
__device__ inline float xb_hist_f64(const XbView &v, double x)
{
    long long i = __double2ll_rn(__dmul_rn(x, (double)v.gc_nbins));
    i = i < 1 ? 1 : (i > v.gc_nbins ? v.gc_nbins : i);
    return v.gc_bins[i - 1];
}
// compute_transcript_bias! (bias.jl:834-858): a thread per base of every transcript
__global__ void xb_bias_kernel(XbView v, int64_t total)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int32_t lo = 0, hi = v.n;  // transcript of base g: last j with tseq_ptr[j] <= g
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (v.tseq_ptr[mid] <= g) lo = mid; else hi = mid;
    }
    const int32_t j = lo;
    const int64_t off = v.tseq_ptr[j], tlen = v.tseq_ptr[j + 1] - off, pos = g - off + 1;
    const uint8_t *seq = v.tseq + off;
    const float sb = xb_seqbias_eval<false>(v, seq, tlen, pos);
    float l;
    if (v.pos_terms) {  // evaluate(posmodel, tlen, tlen - pos + 1) in Float64 (bias.jl:649-658), times the Float32 sequence bias
        const double base = __dadd_rn(__dmul_rn(__ddiv_rn(1.0, (double)tlen), pow(1.0 - v.pos_p, (double)tlen)), v.pos_terms[tlen - 1]);
        const double prob = __dsub_rn(base, v.pos_terms[(tlen - pos + 1) - 1]);
        l = (float)__dmul_rn(__ddiv_rn(prob, base), (double)sb);
    } else {
        l = sb;
    }
    v.left_bias[g] = l;
    v.right_bias[g] = xb_seqbias_eval<true>(v, seq, tlen, pos);
}
// effective_length(::BiasedFragModel, t) (fragmodel.jl:372-410), the inner sum c of one (transcript, fragment length): a
// Float32 sum in sequence over the positions, the GC proportion of the sliding window updated in sequence like the
// reference's.  A thread per pair; csum [n][num_fraglens].
__global__ void xb_efflen_biased_c_kernel(XbView v, float *__restrict__ csum)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)v.n * v.num_fraglens) return;
    const int32_t j = (int32_t)(t / v.num_fraglens), f = (int32_t)(t - (int64_t)j * v.num_fraglens);
    const int64_t off = v.tseq_ptr[j], tlen = v.tseq_ptr[j + 1] - off;
    const int64_t fraglen = v.high_prob_fraglens[f];
    if (fraglen > tlen) {
        csum[t] = 0.0f;
        return;
    }
    const uint8_t *seq = v.tseq + off;
    const float *left = v.left_bias + off, *right = v.right_bias + off;
    const float gc_c = __fdiv_rn(1.0f, (float)fraglen);
    float frag_gc_prop = 0.0f;
    for (int64_t pos = 1; pos <= fraglen; ++pos) frag_gc_prop = __fadd_rn(frag_gc_prop, xb_is_gc(seq[pos - 1]) ? gc_c : 0.0f);
    float c = 0.0f;
    for (int64_t pos = 1; pos <= tlen - fraglen + 1; ++pos) {
        if (pos > 1) {
            frag_gc_prop = __fsub_rn(frag_gc_prop, xb_is_gc(seq[pos - 2]) ? gc_c : 0.0f);
            frag_gc_prop = __fadd_rn(frag_gc_prop, xb_is_gc(seq[pos + fraglen - 2]) ? gc_c : 0.0f);
        }
        c = __fadd_rn(c, __fmul_rn(__fmul_rn(left[pos - 1], right[pos + fraglen - 2]), xb_hist_f32(v, frag_gc_prop)));
    }
    csum[t] = c;
}
// ... and the sum over the fragment lengths, in the list's order (a thread per transcript)
__global__ void xb_efflen_biased_sum_kernel(XbView v, const float *__restrict__ csum, float *__restrict__ efflens)
{
    const int32_t j = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= v.n) return;
    const int64_t tlen = v.tseq_ptr[j + 1] - v.tseq_ptr[j];
    float efflen = 0.0f;
    for (int32_t f = 0; f < v.num_fraglens; ++f) {
        const int64_t fraglen = v.high_prob_fraglens[f];
        if (fraglen > tlen) continue;
        const float fraglenpr = fraglen <= XB_MAX_FRAG_LEN ? v.pmf[fraglen - 1] : 0.0f;
        efflen = __fadd_rn(efflen, __fmul_rn(csum[(int64_t)j * v.num_fraglens + f], fraglenpr));
    }
    efflens[j] = fmaxf(efflen, XB_MIN_EFFECTIVE_LENGTH);
}
// genomic_to_transcriptomic(t, position) (transcripts.jl:520-538): 0 = not in an exon
__device__ inline int64_t xb_g2t_pos(const XbView &v, int32_t j, int64_t position)
{
    const int64_t e0 = v.exon_ptr[j], ne = v.exon_ptr[j + 1] - e0;
    const int64_t *ef = v.exon_first + e0, *el = v.exon_last + e0;
    int64_t lo = 0, hi = ne;  // searchsortedlast(exons, Exon(position, position))
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ef[mid] < position || (ef[mid] == position && el[mid] <= position)) lo = mid + 1; else hi = mid;
    }
    const int64_t i = lo;
    if (i == 0 || el[i - 1] < position) return 0;
    int64_t tpos = 1;
    for (int64_t k = 0; k < i - 1; ++k) tpos += el[k] - ef[k] + 1;
    tpos += position - ef[i - 1];
    if (v.t_strand[j] < 0) tpos = exonic_length(v, j) - tpos + 1;
    return tpos;
}
// condfragprob(::BiasedFragModel, ...) (fragmodel.jl:413-445) with genomic_to_transcriptomic (transcripts.jl:452-517)
__device__ inline float condfragprob_biased(const XbView &v, int32_t j, int64_t i)
{
    int64_t fraglen = fragmentlength(v, j, i);
    if (fraglen < 0) return 0.0f;
    if (fraglen <= 0) {
        fraglen = v.fraglen_median;
        if (fraglen <= 0) return 0.0f;
    }
    const int64_t off = v.tseq_ptr[j], tlen = v.tseq_ptr[j + 1] - off;
    int64_t tpos;
    if (v.m2_left[i] > 0) {
        const int64_t lmin = min(v.m1_left[i], v.m2_left[i]), rmax = max(v.m1_right[i], v.m2_right[i]);
        tpos = xb_g2t_pos(v, j, v.t_strand[j] > 0 ? lmin : rmax);
    } else {
        const bool aln_neg = v.m1_reverse[i] != 0;
        if (v.t_strand[j] > 0) tpos = !aln_neg ? xb_g2t_pos(v, j, v.m1_left[i]) : xb_g2t_pos(v, j, v.m1_right[i]) - fraglen;
        else tpos = !aln_neg ? xb_g2t_pos(v, j, v.m1_left[i]) - fraglen : xb_g2t_pos(v, j, v.m1_right[i]);
    }
    if (tpos <= 0) {
        fraglen += tpos - 1;
        tpos = 1;
    }
    if (tpos + fraglen - 1 > tlen) fraglen = tlen - tpos + 1;
    if (fraglen <= 0) return 0.0f;
    const int64_t a = tpos, b = tpos + fraglen - 1;
    const float fraglenpr = fraglen <= XB_MAX_FRAG_LEN ? v.pmf[fraglen - 1] : 0.0f;
    const uint8_t *seq = v.tseq + off;
    int64_t gc = 0;
    for (int64_t pos = a; pos <= b; ++pos) gc += xb_is_gc(seq[pos - 1]);
    const double frag_gc = __ddiv_rn((double)gc, (double)fraglen);
    const float fragbias = __fmul_rn(__fmul_rn(v.left_bias[off + a - 1], v.right_bias[off + b - 1]), xb_hist_f64(v, frag_gc));
    const float efflen = v.efflens[j];
    if (v.f_strand[i] == v.t_strand[j])
        return __fdiv_rn(__fmul_rn(__fmul_rn(v.strand_specificity, fraglenpr), fragbias), efflen);
    return (float)__ddiv_rn(__dmul_rn(__dmul_rn(1.0 - (double)v.strand_specificity, (double)fraglenpr), (double)fragbias), (double)efflen);
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
        const float p = v.tseq ? condfragprob_biased(v, j, i) : condfragprob(v, j, i);
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
    DevBuf<float> d_efflens, d_vals, d_left_bias, d_right_bias;  // (bias vectors: BiasedFragModel only)
    int64_t total_bases = 0;
    double ms_bias = 0;
    DevBuf<uint32_t> d_cols;
    DevBuf<uint64_t> d_tcolptr;
    DevBuf<int64_t> d_row_fragment;
    double ms_efflen = 0, ms_count = 0, ms_fill = 0;
};

namespace polee {
polee_status xbuild_device_view(const polee_xbuild *xb, polee_ctx **ctx, int64_t *rows, int64_t *n, const uint64_t **tcolptr,
                                const uint32_t **trowval, const float **tnzval)
{
    if (!xb) return fail(nullptr, POLEE_ERR_BAD_ARG, "null xbuild handle");
    *ctx = xb->ctx;
    *rows = xb->rows;
    *n = xb->n;
    *tcolptr = xb->d_tcolptr.p;
    *trowval = xb->d_cols.p;
    *tnzval = xb->d_vals.p;
    return POLEE_OK;
}
}  // namespace polee

extern "C" {

static polee_status polee_xbuild_run_impl(polee_ctx *ctx, const polee_xb_transcripts *T, const polee_xb_fragments *F, const polee_xb_fragmodel *M,
                              const polee_xb_biasmodel *B, polee_xbuild **out)
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

    // the trained bias model (BiasedFragModel): tables present and consistent with the transcripts
    int64_t total_bases = 0;
    if (B) {
        if (!B->tseq_ptr || !B->tseq || !B->orders_left || !B->orders_right || !B->ps_left || !B->ps_right || !B->gc_bins ||
            !B->high_prob_fraglens)
            return fail(ctx, POLEE_ERR_BAD_ARG, "polee_xbuild_run_biased: a bias-model array is null");
        if (B->seqbias_len != XB_BIAS_SEQ_INNER_CTX + XB_BIAS_SEQ_OUTER_CTX)
            return fail(ctx, POLEE_ERR_BAD_ARG, "seqbias_len must be %d (BIAS_SEQ_OUTER_CTX + BIAS_SEQ_INNER_CTX)", XB_BIAS_SEQ_INNER_CTX + XB_BIAS_SEQ_OUTER_CTX);
        if (B->gc_nbins < 1 || B->num_fraglens < 1 || B->ps_ctx < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bias model: empty table");
        int max_order = 0;
        for (int32_t i = 0; i < B->seqbias_len; ++i) max_order = std::max(max_order, std::max(B->orders_left[i], B->orders_right[i]));
        if (max_order > 8 || (int64_t)B->ps_ctx < ((int64_t)1 << (2 * max_order)))
            return fail(ctx, POLEE_ERR_BAD_ARG, "bias model: ps_ctx = %d is smaller than 4^(largest order %d)", B->ps_ctx, max_order);
        if (B->tseq_ptr[0] != 0) return fail(ctx, POLEE_ERR_BAD_ARG, "tseq_ptr[0] must be 0");
        int64_t max_tlen = 0;
        for (int32_t j = 0; j < n; ++j) {
            const int64_t tlen = B->tseq_ptr[j + 1] - B->tseq_ptr[j];
            int64_t ex = 0;
            for (int64_t k = T->exon_ptr[j]; k < T->exon_ptr[j + 1]; ++k) ex += T->exon_last[k] - T->exon_first[k] + 1;
            if (tlen != ex) return fail(ctx, POLEE_ERR_BAD_ARG, "transcript %d: sequence of %lld bases, exons of %lld", j, (long long)tlen, (long long)ex);
            max_tlen = std::max(max_tlen, tlen);
        }
        total_bases = B->tseq_ptr[n];
        if (B->pos_terms && B->pos_maxtlen < max_tlen)
            return fail(ctx, POLEE_ERR_BAD_ARG, "positional bias terms cover transcripts up to %d bases, the longest has %lld", B->pos_maxtlen, (long long)max_tlen);
        for (int32_t f = 0; f < B->num_fraglens; ++f)
            if (B->high_prob_fraglens[f] < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "high_prob_fraglens[%d] = %d", f, B->high_prob_fraglens[f]);
        bool any_single = false;
        for (int64_t i = 0; i < m && !any_single; ++i) any_single = F->m2_left[i] == 0;
        if (any_single && !B->m1_reverse) return fail(ctx, POLEE_ERR_BAD_ARG, "single-end fragments need m1_reverse (transcripts.jl:486)");
    }
    polee_xbuild *xb = new (std::nothrow) polee_xbuild();
    if (!xb) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    xb->total_bases = total_bases;
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
    DevBuf<int64_t> d_tsp;
    DevBuf<uint8_t> d_tsq, d_m1rev;
    DevBuf<int32_t> d_ol, d_or, d_hpf;
    DevBuf<float> d_psl, d_psr, d_gcb, d_csum;
    DevBuf<double> d_pterms;
    if (B) {
        const size_t psn = (size_t)B->seqbias_len * 4 * (size_t)B->ps_ctx;
        std::vector<uint8_t> rev_zero;
        const uint8_t *rev = B->m1_reverse;
        if (!rev) {  // (no single-end fragment: never read)
            rev_zero.assign((size_t)std::max<int64_t>(m, 1), 0);
            rev = rev_zero.data();
        }
        A(d_tsp.upload(ctx, B->tseq_ptr, (size_t)n + 1)); A(d_tsq.upload(ctx, B->tseq, (size_t)std::max<int64_t>(total_bases, 1)));
        A(d_ol.upload(ctx, B->orders_left, (size_t)B->seqbias_len)); A(d_or.upload(ctx, B->orders_right, (size_t)B->seqbias_len));
        A(d_psl.upload(ctx, B->ps_left, psn)); A(d_psr.upload(ctx, B->ps_right, psn));
        A(d_gcb.upload(ctx, B->gc_bins, (size_t)B->gc_nbins)); A(d_hpf.upload(ctx, B->high_prob_fraglens, (size_t)B->num_fraglens));
        if (B->pos_terms) A(d_pterms.upload(ctx, B->pos_terms, (size_t)B->pos_maxtlen));
        A(d_m1rev.upload(ctx, rev, (size_t)std::max<int64_t>(m, 1)));
        A(xb->d_left_bias.alloc(ctx, (size_t)std::max<int64_t>(total_bases, 1))); A(xb->d_right_bias.alloc(ctx, (size_t)std::max<int64_t>(total_bases, 1)));
        A(d_csum.alloc(ctx, (size_t)n * (size_t)B->num_fraglens));
        if (s != POLEE_OK) {
            polee_xbuild_destroy(xb);
            return s;
        }
    }
    XbView v{n, d_tseq.p, d_tstrand.p, d_eptr.p, d_ef.p, d_el.p, d_ord.p, d_ofirst.p, d_omax.p, d_seqptr.p, num_seq, m, d_fseq.p, d_fstrand.p,
             d_m1l.p, d_m1r.p, d_m2l.p, d_m2r.p, d_flag16.p, d_c1.p, d_c2.p, d_cigop.p, d_ciglen.p, d_pmf.p, d_cdf.p, M->fraglen_median,
             M->strand_specificity, M->alt_frag_model, xb->d_efflens.p,
             B ? d_tsp.p : nullptr, B ? d_tsq.p : nullptr, B ? B->seqbias_len : 0, B ? B->ps_ctx : 0, d_ol.p, d_or.p, d_psl.p, d_psr.p,
             B ? B->gc_nbins : 0, d_gcb.p, B ? B->pos_p : 0.0, B && B->pos_terms ? d_pterms.p : nullptr, B ? B->num_fraglens : 0, d_hpf.p,
             d_m1rev.p, xb->d_left_bias.p, xb->d_right_bias.p};
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
    if (B && (s = timed(xb->ms_bias, [&] {
            if (total_bases > 0) hipLaunchKernelGGL(xb_bias_kernel, dim3((unsigned)ceil_div(total_bases, 256)), dim3(256), 0, st, v, total_bases);
        }))) {
        polee_xbuild_destroy(xb);
        return s;
    }
    if ((s = timed(xb->ms_efflen, [&] {
            if (B) {
                hipLaunchKernelGGL(xb_efflen_biased_c_kernel, dim3((unsigned)ceil_div((int64_t)n * B->num_fraglens, 64)), dim3(64), 0, st, v, d_csum.p);
                hipLaunchKernelGGL(xb_efflen_biased_sum_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, st, v, (const float *)d_csum.p, xb->d_efflens.p);
            } else {
                hipLaunchKernelGGL(xb_efflen_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, st, v, xb->d_efflens.p);
            }
        })) ||
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
    return guarded(ctx, "polee_xbuild_run", [&] { return polee_xbuild_run_impl(ctx, T, F, M, nullptr, out); });
}

polee_status polee_xbuild_run_biased(polee_ctx *ctx, const polee_xb_transcripts *T, const polee_xb_fragments *F, const polee_xb_fragmodel *M,
                                     const polee_xb_biasmodel *B, polee_xbuild **out)
{
    if (!B) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_xbuild_run_biased: null bias model");
    return guarded(ctx, "polee_xbuild_run_biased", [&] { return polee_xbuild_run_impl(ctx, T, F, M, B, out); });
}

polee_status polee_xbuild_get_bias(const polee_xbuild *xb, float *left_bias, float *right_bias, double *ms_bias)
{
    if (!xb) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    polee_ctx *ctx = xb->ctx;
    POLEE_TRY(use_device(ctx));
    if (!xb->d_left_bias.p) return fail(ctx, POLEE_ERR_BAD_ARG, "this matrix was built without a bias model");
    if (left_bias) POLEE_TRY(xb->d_left_bias.download(ctx, left_bias, (size_t)xb->total_bases));
    if (right_bias) POLEE_TRY(xb->d_right_bias.download(ctx, right_bias, (size_t)xb->total_bases));
    if (ms_bias) *ms_bias = xb->ms_bias;
    return POLEE_OK;
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
