/*
 * synth.c -- seeded synthetic RNA-Seq sample generator for benchmarks and full-size tests
 * (SURVEY.md 8(d)).  BENCH/TEST SUPPORT, not part of the product library.
 *
 * Model: G = n/3.4 genes laid out along the genome; isoform counts 1+Geometric truncated to
 * [1,30] summing to n; transcript lengths LogNormal(log 1500, 0.8) clipped to [200, 2e4];
 * effective_lengths = max(len - 200, 1) (MIN_EFFECTIVE_LENGTH, src/constants.jl:41); gene
 * abundance LogNormal(0, 2); every gene owns a small set of compatibility patterns (random
 * non-empty subsets of its isoforms -- what exon structure induces in real data), a fragment
 * draws a gene with probability ~ abundance x total length x isoforms^gamma (gamma is solved
 * so that the MEAN nnz per fragment hits the target), then one of the gene's patterns, and
 * with probability 0.05 one extra isoform of the neighbouring gene.
 * Set diversity (how many fragments share a transcript set; synth_set_diversity): `dropout` p drops every entry of a
 * fragment but its first independently with probability p, so that fragments of a gene stop sharing a handful of sets;
 * `literal` = SURVEY 8(d)'s literal wording, every fragment draws its OWN random non-empty subset of its gene's isoforms
 * (each isoform with probability 0.75) instead of one of the gene's <= 12 patterns.  Both off: the stream of random
 * numbers, hence the sample, is the one of rounds 1-2.
 * X_ij = LogNormal(0,1)/efflen_j clipped to [1e-12 (MIN_FRAG_PROB, constants.jl:45), 1e-3].
 * Fragments are emitted in gene (= genomic) order, as src/rnaseq_sample.jl:399-419 produces
 * them, with no empty rows.  Output is Xt (CSR of X): tcolptr u64 [m+1] 1-based, trowval u32
 * 1-based transcript ids (ascending within a row), tnzval f32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r)
{
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double rng_unif(rng_t *r) { return ((double)(rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static inline double rng_norm(rng_t *r)
{
    double u1 = rng_unif(r), u2 = rng_unif(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

#define MAX_ISO 30
#define MAX_PAT 12

typedef struct {
    int32_t first_tid;  /* 0-based id of the gene's first transcript */
    int32_t niso;
    int32_t npat;
    uint32_t pat_mask[MAX_PAT];
    float pat_cum[MAX_PAT]; /* cumulative pattern probabilities */
    double weight;          /* selection weight */
    double mean_pat;        /* expected pattern size */
    int64_t frag_begin, frag_end;
} gene_t;

typedef struct {
    int64_t n, m, nnz;
    int32_t G;
    uint64_t seed;
    gene_t *genes;
    float *efflens;
    double gamma, mean_nnz;
    double dropout; /* per-entry dropout probability (first entry kept) */
    int literal;    /* every fragment draws its own subset of the gene's isoforms */
} synth_t;

static double mean_nnz_for(const synth_t *s, const double *base, double gamma, double *w)
{
    double tot = 0, acc = 0;
    for (int g = 0; g < s->G; ++g) {
        w[g] = base[g] * pow((double)s->genes[g].niso, gamma);
        tot += w[g];
    }
    for (int g = 0; g < s->G; ++g) acc += w[g] / tot * (s->genes[g].mean_pat + 0.05);
    return acc;
}

synth_t *synth_create(int64_t n, int64_t m, double target_nnz, uint64_t seed)
{
    synth_t *s = calloc(1, sizeof(*s));
    s->n = n; s->m = m; s->seed = seed;
    rng_t r = {seed};
    /* isoform counts */
    int32_t *cnt = malloc(sizeof(int32_t) * (n + 1));
    int32_t G = 0; int64_t used = 0;
    const double p = 1.0 / 3.4; /* geometric with mean 1/p - ... on {1,2,..}: mean 3.4 */
    while (used < n) {
        int c = 1 + (int)floor(log(rng_unif(&r)) / log(1.0 - p));
        if (c > MAX_ISO) c = MAX_ISO;
        if (used + c > n) c = (int)(n - used);
        cnt[G++] = c; used += c;
    }
    s->G = G;
    s->genes = calloc(G, sizeof(gene_t));
    s->efflens = malloc(sizeof(float) * n);
    double *base = malloc(sizeof(double) * G), *w = malloc(sizeof(double) * G);
    int32_t tid = 0;
    for (int g = 0; g < G; ++g) {
        gene_t *ge = &s->genes[g];
        ge->first_tid = tid; ge->niso = cnt[g];
        double totlen = 0;
        for (int i = 0; i < cnt[g]; ++i) {
            double len = exp(log(1500.0) + 0.8 * rng_norm(&r));
            len = len < 200 ? 200 : (len > 2e4 ? 2e4 : len);
            double el = len - 200.0; if (el < 1.0) el = 1.0;
            s->efflens[tid + i] = (float)el;
            totlen += len;
        }
        double abundance = exp(2.0 * rng_norm(&r));
        base[g] = abundance * totlen;
        /* compatibility patterns */
        int np = cnt[g] == 1 ? 1 : (2 * cnt[g] < MAX_PAT ? 2 * cnt[g] : MAX_PAT);
        ge->npat = np;
        double psum = 0, pw[MAX_PAT]; ge->mean_pat = 0;
        for (int k = 0; k < np; ++k) {
            uint32_t mask = 0;
            for (int i = 0; i < cnt[g]; ++i) if (rng_unif(&r) < 0.75) mask |= 1u << i;
            if (!mask) mask = 1u << (rng_next(&r) % cnt[g]);
            ge->pat_mask[k] = mask;
            pw[k] = -log(rng_unif(&r)); psum += pw[k];
        }
        double c = 0;
        for (int k = 0; k < np; ++k) {
            c += pw[k] / psum; ge->pat_cum[k] = (float)c;
            ge->mean_pat += pw[k] / psum * __builtin_popcount(ge->pat_mask[k]);
        }
        ge->pat_cum[np - 1] = 1.0f;
        tid += cnt[g];
    }
    /* solve gamma for the target mean nnz (monotone in gamma) */
    double lo = -4, hi = 8;
    for (int it = 0; it < 60; ++it) {
        double mid = 0.5 * (lo + hi);
        if (mean_nnz_for(s, base, mid, w) < target_nnz) lo = mid; else hi = mid;
    }
    s->gamma = 0.5 * (lo + hi);
    s->mean_nnz = mean_nnz_for(s, base, s->gamma, w);
    /* fragments per gene by cumulative rounding; every fragment belongs to exactly one gene */
    double tot = 0, cum = 0;
    for (int g = 0; g < G; ++g) tot += w[g];
    int64_t prev = 0;
    for (int g = 0; g < G; ++g) {
        cum += w[g] / tot;
        int64_t end = g == G - 1 ? m : (int64_t)llround(cum * (double)m);
        if (end < prev) end = prev; if (end > m) end = m;
        s->genes[g].frag_begin = prev; s->genes[g].frag_end = end; s->genes[g].weight = w[g] / tot;
        prev = end;
    }
    free(cnt); free(base); free(w);
    return s;
}

static inline int gen_fragment(const synth_t *s, int g, rng_t *r, uint32_t *cols, float *vals)
{
    const gene_t *ge = &s->genes[g];
    float u = (float)rng_unif(r);
    int k = 0;
    while (k < ge->npat - 1 && u > ge->pat_cum[k]) ++k;
    uint32_t mask = ge->pat_mask[k];
    if (s->literal) {
        mask = 0;
        for (int i = 0; i < ge->niso; ++i) if (rng_unif(r) < 0.75) mask |= 1u << i;
        if (!mask) mask = 1u << (rng_next(r) % ge->niso);
    }
    if (s->dropout > 0.0) {
        int first = 1;
        for (int i = 0; i < ge->niso; ++i)
            if (mask & (1u << i)) {
                if (!first && rng_unif(r) < s->dropout) mask &= ~(1u << i);
                first = 0;
            }
    }
    int cnt = 0;
    int extra = -1;
    if (rng_unif(r) < 0.05 && s->G > 1) {
        int ng = (g + 1 < s->G) ? g + 1 : g - 1;
        extra = s->genes[ng].first_tid + (int)(rng_next(r) % s->genes[ng].niso);
    }
    if (extra >= 0 && extra < ge->first_tid) cols[cnt++] = (uint32_t)extra;
    for (int i = 0; i < ge->niso; ++i) if (mask & (1u << i)) cols[cnt++] = (uint32_t)(ge->first_tid + i);
    if (extra >= 0 && extra > ge->first_tid) cols[cnt++] = (uint32_t)extra;
    if (vals)
        for (int j = 0; j < cnt; ++j) {
            double v = exp(rng_norm(r)) / (double)s->efflens[cols[j]];
            v = v < 1e-12 ? 1e-12 : (v > 1e-3 ? 1e-3 : v);
            vals[j] = (float)v;
        }
    return cnt;
}

static inline rng_t gene_rng(const synth_t *s, int g)
{
    rng_t r = {s->seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(g + 1))};
    rng_next(&r);
    return r;
}

/* pass 1: row lengths -> tcolptr (1-based); returns nnz */
int64_t synth_count(synth_t *s, uint64_t *tcolptr)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int g = 0; g < s->G; ++g) {
        rng_t r = gene_rng(s, g);
        uint32_t cols[MAX_ISO + 2];
        float vals[MAX_ISO + 2];
        for (int64_t i = s->genes[g].frag_begin; i < s->genes[g].frag_end; ++i)
            tcolptr[i + 1] = (uint64_t)gen_fragment(s, g, &r, cols, vals); /* vals drawn to keep the stream aligned */
    }
    tcolptr[0] = 1;
    for (int64_t i = 0; i < s->m; ++i) tcolptr[i + 1] += tcolptr[i];
    s->nnz = (int64_t)(tcolptr[s->m] - 1);
    return s->nnz;
}

/* pass 2: fill (same per-gene streams as pass 1) */
void synth_fill(synth_t *s, const uint64_t *tcolptr, uint32_t *trowval, float *tnzval, float *efflens)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int g = 0; g < s->G; ++g) {
        rng_t r = gene_rng(s, g);
        uint32_t cols[MAX_ISO + 2];
        float vals[MAX_ISO + 2];
        for (int64_t i = s->genes[g].frag_begin; i < s->genes[g].frag_end; ++i) {
            int c = gen_fragment(s, g, &r, cols, vals);
            uint64_t o = tcolptr[i] - 1;
            for (int j = 0; j < c; ++j) { trowval[o + j] = cols[j] + 1; tnzval[o + j] = vals[j]; }
        }
    }
    memcpy(efflens, s->efflens, sizeof(float) * s->n);
}

/* call before synth_count */
void synth_set_diversity(synth_t *s, double dropout, int literal) { s->dropout = dropout; s->literal = literal; }
int32_t synth_num_genes(const synth_t *s) { return s->G; }
double synth_mean_nnz(const synth_t *s) { return s->mean_nnz; }
double synth_gamma(const synth_t *s) { return s->gamma; }
/* gene of each transcript (for building gene-aware trees) */
void synth_gene_of_transcript(const synth_t *s, int32_t *gene)
{
    for (int g = 0; g < s->G; ++g)
        for (int i = 0; i < s->genes[g].niso; ++i) gene[s->genes[g].first_tid + i] = g;
}
void synth_free(synth_t *s) { if (s) { free(s->genes); free(s->efflens); free(s); } }
int synth_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Xt (CSR, 1-based) -> X (CSC, 1-based uint64 colptr) for callers that exercise the CSC entry point */
void synth_csr_to_csc(int64_t m, int64_t n, const uint64_t *tcolptr, const uint32_t *trowval, const float *tnzval,
                      uint64_t *colptr, uint32_t *rowval, float *nzval)
{
    uint64_t nnz = tcolptr[m] - 1;
    memset(colptr, 0, sizeof(uint64_t) * (n + 1));
    for (uint64_t k = 0; k < nnz; ++k) colptr[trowval[k]]++;
    uint64_t run = 1;
    for (int64_t j = 0; j < n; ++j) { uint64_t c = colptr[j + 1]; colptr[j + 1] = run; run += c; }
    colptr[0] = 1; /* colptr[j+1] currently = start of column j; shift while filling */
    uint64_t *cur = malloc(sizeof(uint64_t) * n);
    for (int64_t j = 0; j < n; ++j) cur[j] = colptr[j + 1] - 1;
    for (int64_t i = 0; i < m; ++i)
        for (uint64_t k = tcolptr[i] - 1; k < tcolptr[i + 1] - 1; ++k) {
            uint64_t p = cur[trowval[k] - 1]++;
            rowval[p] = (uint32_t)(i + 1); nzval[p] = tnzval[k];
        }
    for (int64_t j = 0; j < n; ++j) colptr[j] = colptr[j + 1];
    colptr[n] = nnz + 1;
    free(cur);
}
