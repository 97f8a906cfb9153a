/*
 * xbuild_oracle.c -- CPU restatement of the construction of the likelihood matrix X (first
 * slice): which fragments are compatible with which transcripts, and with what conditional probability, under the
 * reference's SimplisticFragModel (bias terms = 1) and, since round 4, its default BiasedFragModel given a trained bias
 * model (second half of this file).
 *
 * TEST INFRASTRUCTURE ONLY: only tests/ may call this (the product path is polee_amd/csrc/xbuild.hip).
 * Parity: UNPINNED -- the reference holds no fixture for this step (its test dataset starts at the likelihood matrix)
 * and Julia cannot run here; every function cites the reference lines it follows.
 *
 *   fragmentlength        src/transcripts.jl:273-446   (CIGAR intervals walked against the transcript's exons / introns)
 *   CigarIter             src/reads.jl:458-492
 *   is_exon_compatible    src/reads.jl:510-518
 *   next exon / intron    src/reads.jl:521-537
 *   effective_length      src/fragmodel.jl:155-169     (SimplisticFragModel)
 *   condfragprob          src/fragmodel.jl:119-153     (SimplisticFragModel)
 *   intersection loop     src/rnaseq_sample.jl:58-121  (pairs (t, alnpr) with alnpr contained in t, fragpr > MIN_FRAG_PROB)
 *   compact_indexes!      src/rnaseq_sample.jl:126-157 (fragments without a compatible transcript are dropped)
 *
 * Conventions of this restatement (inputs are pre-parsed arrays; BAM / GFF parsing is out of scope):
 *   transcripts  j = 0..n-1 (the reference's t.metadata.id - 1), each with a sequence id, a strand (+1 / -1), and its
 *                exons ascending, 1-based inclusive genomic coordinates;
 *   fragments    i = 0..m-1 = alignment pairs: sequence id, strand, the leftmost mate's and (paired-end) the rightmost
 *                mate's alignment as (leftpos, rightpos, CIGAR ops with lengths); `m1_is_flag16` = the lone mate of a
 *                single-end pair has flag == 16 exactly (the reference's `aln.flag == SAM.FLAG_REVERSE != 0`,
 *                fragmodel.jl:130, is a chained comparison);
 *   rows of X    the fragments in input order, those without any entry dropped (the reference numbers them in the order
 *                of its interval trees: a permutation of rows, which the likelihood -- a sum over rows -- does not see).
 * Quirks kept: Float32 accumulation of the effective length in sequence (Float64 from the first term on when
 * alt_frag_model is set and the transcript is longer than MAX_FRAG_LEN, where `denom` is the Float64 literal 1.0);
 * `1.0 - strand_specificity` is Float64.  Quirk NOT kept: searchsortedlast(exons, alnpr) = 0 (an alignment pair that
 * starts at the transcript's first base and ends inside its first exon) indexes exons[0] in the reference -- here the
 * walk starts at the first exon.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_FRAG_LEN 2000            /* src/constants.jl:34 */
#define MIN_EFFECTIVE_LENGTH 1.0f    /* src/constants.jl:41 */
#define MIN_FRAG_PROB 1e-12          /* src/constants.jl:45 */
enum { OP_MATCH = 0, OP_INSERT = 1, OP_DELETE = 2, OP_SKIP = 3, OP_SOFT_CLIP = 4 };  /* BAM operation codes */

typedef struct {
    int32_t n;
    const int32_t *seq;
    const int8_t *strand;
    const int64_t *exon_ptr;  /* [n+1] */
    const int64_t *exon_first, *exon_last;
} xb_transcripts;

typedef struct {
    int64_t m;
    const int32_t *seq;
    const int8_t *strand;
    const int64_t *m1_left, *m1_right;  /* the leftmost mate (or the only one) */
    const int64_t *m2_left, *m2_right;  /* the other mate; m2_left == 0: single-end */
    const uint8_t *m1_is_flag16;
    const int64_t *cig1_ptr, *cig2_ptr; /* [m+1] into cig_op / cig_len; an empty range = one MATCH over [left, right] */
    const uint8_t *cig_op;
    const int32_t *cig_len;
} xb_fragments;

typedef struct {
    const float *fraglen_pmf, *fraglen_cdf; /* [MAX_FRAG_LEN], index l-1 holds length l */
    int32_t fraglen_median;
    float strand_specificity;
    int32_t alt_frag_model;
} xb_fragmodel;

typedef struct { int64_t first, last; int op; } cig_iv;

/* CigarIter (reads.jl:458-492): the i-th interval of an alignment; returns 0 when exhausted */
typedef struct { const uint8_t *op; const int32_t *len; int64_t cnt, i, pos, left, right; } cig_iter;
static int cig_next(cig_iter *it, cig_iv *out)
{
    if (it->cnt == 0) {
        if (it->i > 0) return 0;
        it->i = 1;
        out->first = it->left; out->last = it->right; out->op = OP_MATCH;
        return 1;
    }
    if (it->i >= it->cnt) return 0;
    out->op = it->op[it->i];
    out->first = it->pos;
    out->last = it->pos + it->len[it->i] - 1;
    it->pos += it->len[it->i];
    it->i++;
    return 1;
}

static int exon_compatible(int op) { return op == OP_MATCH || op == OP_SOFT_CLIP || op == OP_INSERT || op == OP_DELETE; }
static int intron_compatible(int op) { return op == OP_SKIP || op == OP_SOFT_CLIP; }

/* reads.jl:521-537 */
static void next_exonintron(const int64_t *ef, const int64_t *el, int64_t ne, int64_t *idx, int *is_exon, int64_t *first, int64_t *last)
{
    if (*is_exon) {
        if (*idx + 1 < ne) { *first = el[*idx] + 1; *last = ef[*idx + 1] - 1; }
        else *idx += 1;
    } else {
        *idx += 1;
        *first = ef[*idx]; *last = el[*idx];
    }
    *is_exon = !*is_exon;
}

/* fragmentlength (transcripts.jl:273-446): -1 = incompatible ("nothing"), 0 = compatible single-end, > 0 = length */
static int64_t fragmentlength(const xb_transcripts *T, const xb_fragments *F, int32_t j, int64_t i)
{
    const int64_t max_enc = 2;
    const int64_t e0 = T->exon_ptr[j], ne = T->exon_ptr[j + 1] - e0;
    const int64_t *ef = T->exon_first + e0, *el = T->exon_last + e0;
    const int paired = F->m2_left[i] > 0;
    const int64_t a_first = F->m1_left[i];
    int64_t a_last = F->m1_right[i];
    if (paired && F->m2_right[i] > a_last) a_last = F->m2_right[i];
    if (a_first < ef[0] || a_last > el[ne - 1]) return -1;  /* :278 */
    /* searchsortedlast(exons, alnpr): last exon <= (first, last) lexicographically; at least the first exon (see header) */
    int64_t first_idx = 0;
    for (int64_t k = 0; k < ne; ++k)
        if (ef[k] < a_first || (ef[k] == a_first && el[k] <= a_last)) first_idx = k; else break;
    cig_iter c1 = {F->cig_op + F->cig1_ptr[i], F->cig_len + F->cig1_ptr[i], F->cig1_ptr[i + 1] - F->cig1_ptr[i], 0, F->m1_left[i], F->m1_left[i], F->m1_right[i]};
    cig_iv c;
    int have = cig_next(&c1, &c);
    int64_t e_idx = first_idx, e_first = ef[e_idx], e_last = el[e_idx];
    int e_isexon = 1;
    int64_t intronlen = 0;
    if (have && c.op == OP_SOFT_CLIP) have = cig_next(&c1, &c);  /* :312 leading soft clipping */
    while (e_idx < ne && have) {
        if (e_last < c.first) {                                                /* case 1 */
            next_exonintron(ef, el, ne, &e_idx, &e_isexon, &e_first, &e_last);
        } else if (c.last >= e_first && c.last <= e_last && c.first >= e_first) { /* case 2 */
            if (e_isexon) { if (!exon_compatible(c.op)) return -1; }
            else { if (!intron_compatible(c.op)) return -1; intronlen += e_last - e_first + 1; }
            have = cig_next(&c1, &c);
        } else if (c.op == OP_SOFT_CLIP) {                                     /* case 3 */
            have = cig_next(&c1, &c);
        } else if (c.last > e_last && c.op == OP_MATCH) {                      /* case 4 */
            if (e_isexon && c.last - e_last <= max_enc) c.last = e_last;
            else if (!e_isexon && e_last >= c.first && e_last - c.first < max_enc) c.first = e_last + 1;
            else return -1;
        } else return -1;                                                      /* case 5 */
    }
    if (have) return -1;
    if (!paired) return 0;                                                     /* :363 */
    int e2_sup_e1 = 0;
    cig_iter c2 = {F->cig_op + F->cig2_ptr[i], F->cig_len + F->cig2_ptr[i], F->cig2_ptr[i + 1] - F->cig2_ptr[i], 0, F->m2_left[i], F->m2_left[i], F->m2_right[i]};
    have = cig_next(&c2, &c);
    int64_t e2_idx = first_idx, e2_first = ef[e2_idx], e2_last = el[e2_idx];
    int e2_isexon = 1;
    while (e2_idx < ne && have) {
        if (e2_last < c.first) {
            if (!e2_isexon && e2_sup_e1) intronlen += e2_last - e2_first + 1;
            if (e_idx < ne && e_first == e2_first && e_last == e2_last) e2_sup_e1 = 1;
            next_exonintron(ef, el, ne, &e2_idx, &e2_isexon, &e2_first, &e2_last);
        } else if (c.last >= e2_first && c.last <= e2_last && c.first >= e2_first) {
            if (e2_isexon) { if (!exon_compatible(c.op)) return -1; }
            else { if (!intron_compatible(c.op)) return -1; }
            have = cig_next(&c2, &c);
        } else if (c.op == OP_SOFT_CLIP) {
            have = cig_next(&c2, &c);
        } else if (c.last > e2_last && c.op == OP_MATCH) {
            if (e2_isexon && c.last - e2_last <= max_enc) c.last = e2_last;
            else if (!e2_isexon && e2_last >= c.first && e2_last - c.first < max_enc) c.first = e2_last + 1;
            else return -1;
        } else return -1;
    }
    if (have && c.op == OP_SOFT_CLIP) have = cig_next(&c2, &c);  /* :430 trailing soft clipping */
    if (have) return -1;
    const int64_t rmax = F->m1_right[i] > F->m2_right[i] ? F->m1_right[i] : F->m2_right[i];
    const int64_t lmin = F->m1_left[i] < F->m2_left[i] ? F->m1_left[i] : F->m2_left[i];
    const int64_t fraglen = rmax - lmin + 1 - intronlen;
    return fraglen > 0 ? fraglen : -1;
}

static int64_t exonic_length(const xb_transcripts *T, int32_t j)
{
    int64_t s = 0;
    for (int64_t k = T->exon_ptr[j]; k < T->exon_ptr[j + 1]; ++k) s += T->exon_last[k] - T->exon_first[k] + 1;
    return s;
}

/* effective_length(::SimplisticFragModel, t) (fragmodel.jl:155-169) */
float xb_oracle_effective_length(const xb_transcripts *T, const xb_fragmodel *M, int32_t j)
{
    const int64_t tlen = exonic_length(T, j);
    const int64_t top = tlen < MAX_FRAG_LEN ? tlen : MAX_FRAG_LEN;
    if (M->alt_frag_model && tlen > MAX_FRAG_LEN) {  /* denom = 1.0 (Float64): el is Float64 from the first term on */
        double el = 0.0;
        for (int64_t l = 1; l <= top; ++l) el += (double)M->fraglen_pmf[l - 1] / 1.0 * (double)(tlen - l + 1);
        const float r = (float)(el > (double)MIN_EFFECTIVE_LENGTH ? el : (double)MIN_EFFECTIVE_LENGTH);
        return r;
    }
    float el = 0.0f;
    if (M->alt_frag_model) {
        const float denom = M->fraglen_cdf[tlen - 1];
        for (int64_t l = 1; l <= top; ++l) el += M->fraglen_pmf[l - 1] / denom * (float)(tlen - l + 1);
    } else {
        for (int64_t l = 1; l <= top; ++l) el += M->fraglen_pmf[l - 1] * (float)(tlen - l + 1);
    }
    return el > MIN_EFFECTIVE_LENGTH ? el : MIN_EFFECTIVE_LENGTH;
}

/* condfragprob(::SimplisticFragModel, ...) (fragmodel.jl:119-153) as the Float32 the reference pushes into V */
float xb_oracle_condfragprob(const xb_transcripts *T, const xb_fragments *F, const xb_fragmodel *M, int32_t j, int64_t i, float efflen)
{
    int64_t fraglen = fragmentlength(T, F, j, i);
    if (fraglen < 0) return 0.0f;
    const int64_t e0 = T->exon_ptr[j], e1 = T->exon_ptr[j + 1];
    if (fraglen <= 0) {  /* single-end read */
        const int64_t maxlen = F->m1_is_flag16[i] ? F->m1_right[i] - T->exon_first[e0] + 1 : T->exon_last[e1 - 1] - F->m1_left[i] + 1;
        fraglen = maxlen < M->fraglen_median ? maxlen : M->fraglen_median;
    }
    const float fraglenpr = fraglen >= 1 && fraglen <= MAX_FRAG_LEN ? M->fraglen_pmf[fraglen - 1] : 0.0f;
    double fragpr;
    if (F->strand[i] == T->strand[j]) fragpr = (double)(M->strand_specificity * fraglenpr / efflen);          /* Float32 chain */
    else fragpr = (1.0 - (double)M->strand_specificity) * (double)fraglenpr / (double)efflen;               /* Float64 chain */
    if (M->alt_frag_model) {
        const int64_t tlen = exonic_length(T, j);
        if (tlen <= MAX_FRAG_LEN) {
            if (F->strand[i] == T->strand[j]) fragpr = (double)((float)fragpr / M->fraglen_cdf[tlen - 1]);
            else fragpr = fragpr / (double)M->fraglen_cdf[tlen - 1];
        }
    }
    return (float)fragpr;
}

/* The whole step: effective lengths, every (fragment, transcript) pair with the fragment's span inside the transcript's
 * (same sequence), entries with finite fragpr > MIN_FRAG_PROB, empty rows dropped.  Outputs are malloc'ed:
 * tcolptr u64 [rows+1] 1-based, trowval u32 1-based ascending within a row, tnzval f32, row_fragment i64 [rows]. */
int xb_oracle_build(const xb_transcripts *T, const xb_fragments *F, const xb_fragmodel *M, float *efflens,
                    int64_t *rows_out, uint64_t **tcolptr, uint32_t **trowval, float **tnzval, int64_t **row_fragment)
{
    for (int32_t j = 0; j < T->n; ++j) efflens[j] = xb_oracle_effective_length(T, M, j);
    size_t cap = 1024, nnz = 0;
    uint32_t *cols = malloc(cap * sizeof(uint32_t));
    float *vals = malloc(cap * sizeof(float));
    uint64_t *ptr = malloc(((size_t)F->m + 2) * sizeof(uint64_t));
    int64_t *rf = malloc(((size_t)F->m + 1) * sizeof(int64_t));
    int64_t rows = 0;
    ptr[0] = 1;
    for (int64_t i = 0; i < F->m; ++i) {
        const size_t start = nnz;
        int64_t a_last = F->m1_right[i];
        if (F->m2_left[i] > 0 && F->m2_right[i] > a_last) a_last = F->m2_right[i];
        for (int32_t j = 0; j < T->n; ++j) {  /* (brute force: the oracle favours obviousness) */
            if (T->seq[j] != F->seq[i]) continue;
            const int64_t tf = T->exon_first[T->exon_ptr[j]], tl = T->exon_last[T->exon_ptr[j + 1] - 1];
            if (!(tf <= F->m1_left[i] && a_last <= tl)) continue;  /* intersect_contains (rnaseq_sample.jl:77-79) */
            const float p = xb_oracle_condfragprob(T, F, M, j, i, efflens[j]);
            if (isfinite(p) && (double)p > MIN_FRAG_PROB) {
                if (nnz == cap) { cap *= 2; cols = realloc(cols, cap * sizeof(uint32_t)); vals = realloc(vals, cap * sizeof(float)); }
                cols[nnz] = (uint32_t)j + 1; vals[nnz] = p; ++nnz;
            }
        }
        if (nnz > start) { rf[rows] = i; ptr[++rows] = (uint64_t)nnz + 1; }
    }
    *rows_out = rows; *tcolptr = ptr; *trowval = cols; *tnzval = vals; *row_fragment = rf;
    return 0;
}

/* ===================================================================================================================
 * BiasedFragModel (round 4; the reference's DEFAULT model, main.jl:689,703): evaluation of a TRAINED bias model during the
 * construction of X.  Training (bias.jl:266-400, 464-515, 532-648, 677-786) stays upstream, like BAM parsing.
 *
 *   evaluate(::SeqBiasModel{:left / :right}, seq, pos)   src/bias.jl:419-456
 *   evaluate(::SimpleHistogramModel, x)                  src/bias.jl:517-520
 *   evaluate(::PositionalBiasModel, tlen, pos)           src/bias.jl:649-663
 *   compute_transcript_bias!                             src/bias.jl:834-858
 *   effective_length(::BiasedFragModel, t)               src/fragmodel.jl:372-410
 *   genomic_to_transcriptomic                            src/transcripts.jl:452-538
 *   condfragprob(::BiasedFragModel, ...)                 src/fragmodel.jl:413-445
 *
 * Quirk NOT reproducible: context positions beyond a transcript's ends are random nucleotides in the reference
 * (randdna(), bias.jl:83-85 in :424-429,444-449) -- a run-to-run random quantity; here they read as A (code 0).
 * Quirks kept: Float32 running GC proportion of the effective length's sliding window (+= / -= in sequence), Float32
 * products left to right, `round(Int, x * nbins)` to even, the single-end position arithmetic as written (:486-501),
 * the overhang nudges (:505-512), `1.0 - strand_specificity` in Float64.
 */
#define BIAS_SEQ_INNER_CTX 15 /* src/constants.jl:77 */
#define BIAS_SEQ_OUTER_CTX 5  /* src/constants.jl:78 */
typedef struct {
    const int64_t *tseq_ptr;  /* [n+1] into tseq */
    const uint8_t *tseq;      /* spliced sequence in transcript orientation: 0 A, 1 C, 2 G, 3 T, 4 anything else */
    int32_t seqbias_len;      /* BIAS_SEQ_OUTER_CTX + BIAS_SEQ_INNER_CTX */
    int32_t ps_ctx;           /* 4^(largest order) */
    const int32_t *orders_left, *orders_right; /* [seqbias_len], -1 = position not in the model */
    const float *ps_left, *ps_right;           /* [seqbias_len][4][ps_ctx] */
    int32_t gc_nbins;
    const float *gc_bins;
    double pos_p;
    const double *pos_terms;  /* [pos_maxtlen] or NULL (pos_model === nothing) */
    int32_t pos_maxtlen;
    int32_t num_fraglens;
    const int32_t *high_prob_fraglens;
    const uint8_t *m1_reverse; /* [m] the lone mate's flag has FLAG_REVERSE set (transcripts.jl:486) */
} xb_biasmodel;

static int code_at(const uint8_t *seq, int64_t len, int64_t j) /* 1-based; off the ends: A (see header) */
{
    if (j < 1 || j > len) return 0;
    return seq[j - 1] < 4 ? seq[j - 1] : 0; /* nt2bit: N -> 0 (bias.jl:176-181) */
}
static int is_gc(uint8_t c) { return c == 1 || c == 2; }

/* evaluate(sb, seq, pos): first = pos - OUTER (left) / pos - INNER + 1 (right) */
static float seqbias_eval(const xb_biasmodel *B, int right, const uint8_t *seq, int64_t len, int64_t pos)
{
    const int32_t *orders = right ? B->orders_right : B->orders_left;
    const float *ps = right ? B->ps_right : B->ps_left;
    const int64_t first = right ? pos - BIAS_SEQ_INNER_CTX + 1 : pos - BIAS_SEQ_OUTER_CTX;
    float bias = 1.0f;
    for (int32_t i = 0; i < B->seqbias_len; ++i) {
        if (orders[i] < 0) continue;
        const int64_t j = first + i;
        const int c = code_at(seq, len, j);
        int ctx = 0;
        for (int l = 1; l <= orders[i]; ++l) ctx = (ctx << 2) | code_at(seq, len, j + l);
        bias *= ps[((size_t)i * 4 + (size_t)c) * (size_t)B->ps_ctx + (size_t)ctx];
    }
    return bias;
}
static float hist_eval_f32(const xb_biasmodel *B, float x)
{
    long i = lrintf(x * (float)B->gc_nbins); /* round(Int, x * length(bins)): Float32 product, ties to even */
    if (i < 1) i = 1;
    if (i > B->gc_nbins) i = B->gc_nbins;
    return B->gc_bins[i - 1];
}
static float hist_eval_f64(const xb_biasmodel *B, double x)
{
    long i = lrint(x * (double)B->gc_nbins);
    if (i < 1) i = 1;
    if (i > B->gc_nbins) i = B->gc_nbins;
    return B->gc_bins[i - 1];
}
/* compute_transcript_bias! (bias.jl:834-858): left / right [tlen] */
void xb_oracle_transcript_bias(const xb_biasmodel *B, int32_t j, float *left, float *right)
{
    const uint8_t *seq = B->tseq + B->tseq_ptr[j];
    const int64_t tlen = B->tseq_ptr[j + 1] - B->tseq_ptr[j];
    for (int64_t pos = 1; pos <= tlen; ++pos) {
        const float sb = seqbias_eval(B, 0, seq, tlen, pos);
        if (B->pos_terms) { /* evaluate(posmodel, tlen, tlen - pos + 1), Float64 (bias.jl:649-658) */
            const double base = (1.0 / (double)tlen) * pow(1.0 - B->pos_p, (double)tlen) + B->pos_terms[tlen - 1];
            const double prob = base - B->pos_terms[(tlen - pos + 1) - 1];
            left[pos - 1] = (float)((prob / base) * (double)sb);
        } else {
            left[pos - 1] = 1.0f * sb;
        }
        right[pos - 1] = seqbias_eval(B, 1, seq, tlen, pos);
    }
}
/* effective_length(::BiasedFragModel, t) (fragmodel.jl:372-410) */
float xb_oracle_effective_length_biased(const xb_fragmodel *M, const xb_biasmodel *B, int32_t j, const float *left, const float *right)
{
    const uint8_t *seq = B->tseq + B->tseq_ptr[j];
    const int64_t tlen = B->tseq_ptr[j + 1] - B->tseq_ptr[j];
    float efflen = 0.0f;
    for (int32_t f = 0; f < B->num_fraglens; ++f) {
        const int64_t fraglen = B->high_prob_fraglens[f];
        if (fraglen > tlen) continue;
        const float fraglenpr = fraglen <= MAX_FRAG_LEN ? M->fraglen_pmf[fraglen - 1] : 0.0f;
        const float gc_c = 1.0f / (float)fraglen;
        float frag_gc_prop = 0.0f;
        for (int64_t pos = 1; pos <= fraglen; ++pos) frag_gc_prop += gc_c * (float)is_gc(seq[pos - 1]);
        float c = 0.0f;
        for (int64_t pos = 1; pos <= tlen - fraglen + 1; ++pos) {
            if (pos > 1) {
                frag_gc_prop -= gc_c * (float)is_gc(seq[pos - 2]);
                frag_gc_prop += gc_c * (float)is_gc(seq[pos + fraglen - 2]);
            }
            c += left[pos - 1] * right[pos + fraglen - 2] * hist_eval_f32(B, frag_gc_prop);
        }
        efflen += c * fraglenpr;
    }
    return efflen > MIN_EFFECTIVE_LENGTH ? efflen : MIN_EFFECTIVE_LENGTH;
}
/* genomic_to_transcriptomic(t, position) (transcripts.jl:520-538): 0 = not in an exon */
static int64_t g2t_pos(const xb_transcripts *T, int32_t j, int64_t position)
{
    const int64_t e0 = T->exon_ptr[j], ne = T->exon_ptr[j + 1] - e0;
    const int64_t *ef = T->exon_first + e0, *el = T->exon_last + e0;
    int64_t i = 0; /* searchsortedlast(exons, Exon(position, position)): exons <= (position, position) lexicographically */
    for (int64_t k = 0; k < ne; ++k)
        if (ef[k] < position || (ef[k] == position && el[k] <= position)) i = k + 1; else break;
    if (i == 0 || el[i - 1] < position) return 0;
    int64_t tpos = 1;
    for (int64_t k = 0; k < i - 1; ++k) tpos += el[k] - ef[k] + 1;
    tpos += position - ef[i - 1];
    if (T->strand[j] < 0) tpos = exonic_length(T, j) - tpos + 1;
    return tpos;
}
/* genomic_to_transcriptomic(t, rs, alnpr, fraglen_median) (transcripts.jl:452-517): the fragment's interval on the
 * transcript, [*start, *stop]; returns its length (0 = incompatible / empty) */
static int64_t g2t_fragment(const xb_transcripts *T, const xb_fragments *F, const xb_biasmodel *B, int32_t j, int64_t i,
                            int64_t fraglen_median, int64_t tlen, int64_t *start, int64_t *stop)
{
    int64_t fraglen = fragmentlength(T, F, j, i);
    if (fraglen < 0) return 0;          /* nothing: incompatible */
    if (fraglen <= 0) {
        fraglen = fraglen_median;
        if (fraglen <= 0) return 0;
    }
    int64_t tpos;
    if (F->m2_left[i] > 0) {            /* both mates */
        const int64_t lmin = F->m1_left[i] < F->m2_left[i] ? F->m1_left[i] : F->m2_left[i];
        const int64_t rmax = F->m1_right[i] > F->m2_right[i] ? F->m1_right[i] : F->m2_right[i];
        tpos = g2t_pos(T, j, T->strand[j] > 0 ? lmin : rmax);
    } else {                            /* single-end: guess (:481-501) */
        const int aln_neg = B->m1_reverse[i] != 0;
        if (T->strand[j] > 0) tpos = !aln_neg ? g2t_pos(T, j, F->m1_left[i]) : g2t_pos(T, j, F->m1_right[i]) - fraglen;
        else tpos = !aln_neg ? g2t_pos(T, j, F->m1_left[i]) - fraglen : g2t_pos(T, j, F->m1_right[i]);
    }
    if (tpos <= 0) { fraglen += tpos - 1; tpos = 1; }               /* :505-508 */
    if (tpos + fraglen - 1 > tlen) fraglen = tlen - tpos + 1;       /* :510-512 */
    *start = tpos; *stop = tpos + fraglen - 1;
    return fraglen > 0 ? fraglen : 0;
}
/* condfragprob(::BiasedFragModel, ...) (fragmodel.jl:413-445) as the Float32 pushed into V */
float xb_oracle_condfragprob_biased(const xb_transcripts *T, const xb_fragments *F, const xb_fragmodel *M, const xb_biasmodel *B,
                                    int32_t j, int64_t i, float efflen, const float *left, const float *right)
{
    const uint8_t *seq = B->tseq + B->tseq_ptr[j];
    const int64_t tlen = B->tseq_ptr[j + 1] - B->tseq_ptr[j];
    int64_t a, b;
    const int64_t fraglen = g2t_fragment(T, F, B, j, i, M->fraglen_median, tlen, &a, &b);
    if (fraglen == 0) return 0.0f;
    const float fraglenpr = fraglen <= MAX_FRAG_LEN ? M->fraglen_pmf[fraglen - 1] : 0.0f;
    int64_t gc = 0;
    for (int64_t pos = a; pos <= b; ++pos) gc += is_gc(seq[pos - 1]);
    const double frag_gc = (double)gc / (double)fraglen;
    const float fragbias = left[a - 1] * right[b - 1] * hist_eval_f64(B, frag_gc);
    if (F->strand[i] == T->strand[j]) return M->strand_specificity * fraglenpr * fragbias / efflen;  /* Float32 chain */
    return (float)((1.0 - (double)M->strand_specificity) * (double)fraglenpr * (double)fragbias / (double)efflen);
}
/* the whole step under the BiasedFragModel; also returns the bias vectors (left / right, concatenated like tseq) when
 * the pointers are non-NULL (caller-allocated, tseq_ptr[n] floats each) */
int xb_oracle_build_biased(const xb_transcripts *T, const xb_fragments *F, const xb_fragmodel *M, const xb_biasmodel *B,
                           float *efflens, float *left_out, float *right_out, int64_t *rows_out, uint64_t **tcolptr,
                           uint32_t **trowval, float **tnzval, int64_t **row_fragment)
{
    const int64_t total = B->tseq_ptr[T->n];
    float *left = left_out ? left_out : malloc((size_t)(total > 0 ? total : 1) * sizeof(float));
    float *right = right_out ? right_out : malloc((size_t)(total > 0 ? total : 1) * sizeof(float));
    for (int32_t j = 0; j < T->n; ++j) {
        xb_oracle_transcript_bias(B, j, left + B->tseq_ptr[j], right + B->tseq_ptr[j]);
        efflens[j] = xb_oracle_effective_length_biased(M, B, j, left + B->tseq_ptr[j], right + B->tseq_ptr[j]);
    }
    size_t cap = 1024, nnz = 0;
    uint32_t *cols = malloc(cap * sizeof(uint32_t));
    float *vals = malloc(cap * sizeof(float));
    uint64_t *ptr = malloc(((size_t)F->m + 2) * sizeof(uint64_t));
    int64_t *rf = malloc(((size_t)F->m + 1) * sizeof(int64_t));
    int64_t rows = 0;
    ptr[0] = 1;
    for (int64_t i = 0; i < F->m; ++i) {
        const size_t start = nnz;
        int64_t a_last = F->m1_right[i];
        if (F->m2_left[i] > 0 && F->m2_right[i] > a_last) a_last = F->m2_right[i];
        for (int32_t j = 0; j < T->n; ++j) {
            if (T->seq[j] != F->seq[i]) continue;
            const int64_t tf = T->exon_first[T->exon_ptr[j]], tl = T->exon_last[T->exon_ptr[j + 1] - 1];
            if (!(tf <= F->m1_left[i] && a_last <= tl)) continue;
            const float p = xb_oracle_condfragprob_biased(T, F, M, B, j, i, efflens[j], left + B->tseq_ptr[j], right + B->tseq_ptr[j]);
            if (isfinite(p) && (double)p > MIN_FRAG_PROB) {
                if (nnz == cap) { cap *= 2; cols = realloc(cols, cap * sizeof(uint32_t)); vals = realloc(vals, cap * sizeof(float)); }
                cols[nnz] = (uint32_t)j + 1; vals[nnz] = p; ++nnz;
            }
        }
        if (nnz > start) { rf[rows] = i; ptr[++rows] = (uint64_t)nnz + 1; }
    }
    if (!left_out) free(left);
    if (!right_out) free(right);
    *rows_out = rows; *tcolptr = ptr; *trowval = cols; *tnzval = vals; *row_fragment = rf;
    return 0;
}

void xb_oracle_free(void *p) { free(p); }
