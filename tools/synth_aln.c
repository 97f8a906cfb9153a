/*
 * synth_aln.c -- seeded synthetic ALIGNMENTS for the X-construction path: gene models with exons
 * and alternative isoforms on a few sequences, and paired-end / single-end fragments sampled from the transcripts and
 * mapped back to the genome (CIGAR: matches, introns as N, a few soft clips), plus fragments from nowhere.
 * BENCH / TEST SUPPORT, not part of the product library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rnext(rng_t *r)
{
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double runif(rng_t *r) { return ((double)(rnext(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static inline int64_t rint_(rng_t *r, int64_t lo, int64_t hi) { return lo + (int64_t)(rnext(r) % (uint64_t)(hi - lo + 1)); }

typedef struct {
    int32_t n, num_seq;
    int32_t *seq; int8_t *strand; int64_t *exon_ptr, *exon_first, *exon_last;
    int32_t *gene_first_t, *gene_nt; int32_t ngenes; double *gene_cum;
} model_t;

/* genes of 1..6 isoforms over 2..9 exons; returns a model with ~n transcripts (exactly n) */
model_t *aln_model_create(int32_t n, int32_t num_seq, uint64_t seed)
{
    rng_t r = {seed};
    model_t *M = calloc(1, sizeof(*M));
    M->n = n; M->num_seq = num_seq;
    M->seq = malloc(sizeof(int32_t) * n); M->strand = malloc(n);
    M->exon_ptr = malloc(sizeof(int64_t) * (n + 1));
    int64_t cap = (int64_t)n * 9 + 16, ne = 0;
    M->exon_first = malloc(sizeof(int64_t) * cap); M->exon_last = malloc(sizeof(int64_t) * cap);
    M->gene_first_t = malloc(sizeof(int32_t) * n); M->gene_nt = malloc(sizeof(int32_t) * n); M->gene_cum = malloc(sizeof(double) * n);
    int64_t *pos = calloc(num_seq, sizeof(int64_t));
    for (int s = 0; s < num_seq; ++s) pos[s] = 1000;
    int32_t t = 0, g = 0; double cum = 0;
    M->exon_ptr[0] = 0;
    while (t < n) {
        const int s = (int)(rnext(&r) % (uint64_t)num_seq);
        const int8_t strand = (rnext(&r) & 1) ? 1 : -1;
        const int nex = (int)rint_(&r, 2, 9);
        int64_t ef[9], el[9], p = pos[s] + rint_(&r, 200, 4000);
        if (runif(&r) < 0.15) p = pos[s] - rint_(&r, 100, 1500);  /* overlapping genes */
        if (p < 100) p = 100;
        for (int e = 0; e < nex; ++e) { ef[e] = p; el[e] = p + rint_(&r, 60, 700); p = el[e] + 1 + rint_(&r, 80, 2500); }
        if (p > pos[s]) pos[s] = p;
        int niso = (int)rint_(&r, 1, 6);
        if (t + niso > n) niso = n - t;
        M->gene_first_t[g] = t; M->gene_nt[g] = niso;
        for (int k = 0; k < niso; ++k) {
            /* isoform: a non-empty subset of the exons (the first isoform has them all); alternative 3' ends now and then */
            int cnt = 0;
            for (int e = 0; e < nex; ++e) {
                if (k == 0 || runif(&r) < 0.7 || (cnt == 0 && e == nex - 1)) {
                    M->exon_first[ne] = ef[e]; M->exon_last[ne] = el[e];
                    if (k > 0 && runif(&r) < 0.1 && el[e] - ef[e] > 40) M->exon_last[ne] -= rint_(&r, 3, 30);
                    ++ne; ++cnt;
                }
            }
            M->seq[t] = s; M->strand[t] = strand; M->exon_ptr[t + 1] = ne; ++t;
        }
        cum += exp(1.5 * sqrt(-2.0 * log(runif(&r))) * cos(6.283185307179586 * runif(&r)));
        M->gene_cum[g] = cum; ++g;
    }
    M->ngenes = g;
    free(pos);
    return M;
}
int64_t aln_model_num_exons(const model_t *M) { return M->exon_ptr[M->n]; }
void aln_model_get(const model_t *M, int32_t *seq, int8_t *strand, int64_t *exon_ptr, int64_t *ef, int64_t *el)
{
    memcpy(seq, M->seq, sizeof(int32_t) * M->n); memcpy(strand, M->strand, M->n);
    memcpy(exon_ptr, M->exon_ptr, sizeof(int64_t) * (M->n + 1));
    memcpy(ef, M->exon_first, sizeof(int64_t) * M->exon_ptr[M->n]); memcpy(el, M->exon_last, sizeof(int64_t) * M->exon_ptr[M->n]);
}
void aln_model_free(model_t *M)
{
    if (!M) return;
    free(M->seq); free(M->strand); free(M->exon_ptr); free(M->exon_first); free(M->exon_last);
    free(M->gene_first_t); free(M->gene_nt); free(M->gene_cum); free(M);
}

/* transcript interval [a, b] (1-based, in transcript coordinates) -> genomic CIGAR; returns ops written */
static int map_interval(const model_t *M, int32_t t, int64_t a, int64_t b, int64_t *left, int64_t *right, uint8_t *op, int32_t *len)
{
    int nops = 0; int64_t off = 0; int started = 0; int64_t prev_end = 0;
    for (int64_t k = M->exon_ptr[t]; k < M->exon_ptr[t + 1]; ++k) {
        const int64_t elen = M->exon_last[k] - M->exon_first[k] + 1;
        const int64_t lo = a - off > 1 ? a - off : 1, hi = b - off < elen ? b - off : elen;  /* part of [a, b] inside this exon */
        if (lo <= hi) {
            const int64_t gs = M->exon_first[k] + lo - 1, ge = M->exon_first[k] + hi - 1;
            if (!started) { *left = gs; started = 1; }
            else { op[nops] = 3; len[nops] = (int32_t)(gs - prev_end - 1); ++nops; }
            op[nops] = 0; len[nops] = (int32_t)(ge - gs + 1); ++nops;
            prev_end = ge; *right = ge;
        }
        off += elen;
    }
    return nops;
}

/*
 * Fragments: m of them; with probability p_noise a fragment comes from a random place (compatible with nothing, mostly).
 * Outputs sized by the caller: per fragment arrays [m]; cig_ptr [m+1] x 2; cig arrays with room for 24 ops per fragment.
 * Fragments are emitted sorted by (sequence, leftmost position).  Returns the number of CIGAR operations.
 */
int64_t aln_fragments(const model_t *M, int64_t m, uint64_t seed, const float *pmf, int read_len, double p_single, double p_noise,
                      double strand_specificity, int32_t *f_seq, int8_t *f_strand, int64_t *m1l, int64_t *m1r, int64_t *m2l,
                      int64_t *m2r, uint8_t *flag16, int64_t *c1p, int64_t *c2p, uint8_t *cop, int32_t *clen, int32_t *true_t)
{
    rng_t r = {seed ^ 0xABCDEF12345ull};
    float cdf[2000]; double c = 0; for (int l = 0; l < 2000; ++l) { c += pmf[l]; cdf[l] = (float)c; }
    typedef struct { int32_t seq; int8_t strand; int64_t a1l, a1r, a2l, a2r; uint8_t f16; int n1, n2; uint8_t op[24]; int32_t len[24]; int32_t tt; } frag_t;
    frag_t *F = malloc(sizeof(frag_t) * m);
    for (int64_t i = 0; i < m; ++i) {
        frag_t *f = &F[i]; memset(f, 0, sizeof(*f)); f->tt = -1;
        if (runif(&r) < p_noise) {
            f->seq = (int32_t)(rnext(&r) % (uint64_t)M->num_seq); f->strand = (rnext(&r) & 1) ? 1 : -1;
            f->a1l = rint_(&r, 500, 3000000); f->a1r = f->a1l + read_len - 1; f->n1 = 0; f->n2 = 0;
            if (runif(&r) < 0.5) { f->a2l = f->a1l + rint_(&r, 50, 400); f->a2r = f->a2l + read_len - 1; }
            continue;
        }
        /* gene by abundance, isoform uniformly */
        const double u = runif(&r) * M->gene_cum[M->ngenes - 1];
        int lo = 0, hi = M->ngenes - 1;
        while (lo < hi) { int mid = (lo + hi) / 2; if (M->gene_cum[mid] < u) lo = mid + 1; else hi = mid; }
        const int32_t t = M->gene_first_t[lo] + (int32_t)(rnext(&r) % (uint64_t)M->gene_nt[lo]);
        int64_t tlen = 0; for (int64_t k = M->exon_ptr[t]; k < M->exon_ptr[t + 1]; ++k) tlen += M->exon_last[k] - M->exon_first[k] + 1;
        const float uu = (float)runif(&r) * cdf[1999];
        int fl = 0; while (fl < 1999 && cdf[fl] < uu) ++fl; fl += 1;
        if (fl > tlen) fl = (int)tlen;
        const int64_t a = rint_(&r, 1, tlen - fl + 1), b = a + fl - 1;
        const int rl = read_len < fl ? read_len : fl;
        f->seq = M->seq[t]; f->tt = t;
        f->strand = runif(&r) < strand_specificity ? M->strand[t] : (int8_t)-M->strand[t];
        const int single = runif(&r) < p_single;
        f->n1 = map_interval(M, t, a, a + rl - 1, &f->a1l, &f->a1r, f->op, f->len);
        if (!single) f->n2 = map_interval(M, t, b - rl + 1, b, &f->a2l, &f->a2r, f->op + f->n1, f->len + f->n1);
        else { f->a2l = 0; f->a2r = 0; f->f16 = (uint8_t)(rnext(&r) & 1); }
        /* now and then: a soft clip in front of mate 1 (its bases are not aligned: the interval shrinks) */
        if (runif(&r) < 0.05 && f->len[0] > 12 && f->n1 + f->n2 < 22) {
            const int sc = (int)rint_(&r, 1, 8);
            memmove(f->op + 1, f->op, f->n1 + f->n2); memmove(f->len + 1, f->len, sizeof(int32_t) * (f->n1 + f->n2));
            f->op[0] = 4; f->len[0] = sc; f->len[1] -= sc; f->n1 += 1;
            /* (CigarIter starts at leftpos and the clipped bases take coordinates there: leftpos stays, as in the
               reference's reader, where soft-clipped bases are counted into the interval walk and skipped) */
        }
        /* a single match over the mate needs no CIGAR array */
        if (f->n1 == 1 && f->op[0] == 0 && runif(&r) < 0.5) { memmove(f->op, f->op + 1, f->n2); memmove(f->len, f->len + 1, sizeof(int32_t) * f->n2); f->n1 = 0; }
    }
    /* sort by (seq, left) -- insertion into index order via qsort */
    int64_t *idx = malloc(sizeof(int64_t) * m);
    for (int64_t i = 0; i < m; ++i) idx[i] = i;
    /* simple radix-free sort: qsort with a static pointer */
    static frag_t *GF; GF = F;
    int cmp(const void *x, const void *y) {
        const frag_t *a = &GF[*(const int64_t *)x], *b = &GF[*(const int64_t *)y];
        if (a->seq != b->seq) return a->seq < b->seq ? -1 : 1;
        if (a->a1l != b->a1l) return a->a1l < b->a1l ? -1 : 1;
        return *(const int64_t *)x < *(const int64_t *)y ? -1 : 1;
    }
    qsort(idx, m, sizeof(int64_t), cmp);
    /* all first mates' operations, then all second mates': cig1_ptr / cig2_ptr are [m+1] offset arrays each */
    int64_t nc = 0;
    for (int64_t q = 0; q < m; ++q) {
        const frag_t *f = &F[idx[q]];
        f_seq[q] = f->seq; f_strand[q] = f->strand; m1l[q] = f->a1l; m1r[q] = f->a1r; m2l[q] = f->a2l; m2r[q] = f->a2r; flag16[q] = f->f16;
        true_t[q] = f->tt;
        c1p[q] = nc; memcpy(cop + nc, f->op, f->n1); memcpy(clen + nc, f->len, sizeof(int32_t) * f->n1); nc += f->n1;
    }
    c1p[m] = nc;
    for (int64_t q = 0; q < m; ++q) {
        const frag_t *f = &F[idx[q]];
        c2p[q] = nc; memcpy(cop + nc, f->op + f->n1, f->n2); memcpy(clen + nc, f->len + f->n1, sizeof(int32_t) * f->n2); nc += f->n2;
    }
    c2p[m] = nc;
    free(idx); free(F);
    return nc;
}
