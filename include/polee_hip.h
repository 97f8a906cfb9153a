/*
 * polee_hip.h -- C ABI of libpolee_hip.so, the MI355X (gfx950) approximate-likelihood
 * engine for Polee.  This is the drop-in boundary: the reference's Julia host code
 * reaches these symbols with `ccall` (see INTEGRATION.md and julia/PoleeHIP.jl); they
 * replace the PyCall + TensorFlow + hsb_ops.so path entirely.
 *
 * Conventions
 *   - Every function returns a polee_status (0 = ok).  A message for the last failure
 *     on a context is available from polee_last_error(ctx); failures that happen
 *     before a context exists are reported by polee_last_error(NULL).
 *   - Host pointers are borrowed for the duration of the call only (Julia callers
 *     must GC.@preserve them).  The library copies what it needs to the device.
 *     Pointers named d_* are DEVICE pointers.
 *   - Index arrays use the reference's on-disk conventions: node_parent_idxs, node_js,
 *     colptr, rowval are 1-based exactly as in the prep / likelihood-matrix HDF5 files;
 *     left/right/leaf_index are 0-based with -1 = none as make_inverse_ptt_params
 *     (src/ptt.jl:293-309) produces them for the TF ops.
 *   - Batched arrays are row-major [B][len].
 *   - One HIP stream per context; a handle must not be used from two host threads at
 *     once; different contexts (different GPUs) may be driven concurrently.  The
 *     library calls hipSetDevice itself on every entry (safe from any OS thread).
 *   - Nothing here falls back to the CPU: without a usable GPU polee_ctx_create fails.
 *
 * Citations are reference file:line, relative to the reference repository root.
 */
#ifndef POLEE_HIP_H
#define POLEE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int polee_status;
enum {
    POLEE_OK = 0,
    POLEE_ERR_BAD_ARG = 1,     /* size / NULL / malformed tree or matrix                  */
    POLEE_ERR_OOM = 2,         /* host or device allocation failed                        */
    POLEE_ERR_HIP = 3,         /* a HIP runtime call or kernel launch failed              */
    POLEE_ERR_NONFINITE = 4,   /* non-finite gradient / likelihood (mirrors the @assert   */
                               /* isfinite at likelihood-approximation.jl:559,564,        */
                               /* likelihood.jl:50, ptt.jl:156-157)                       */
    POLEE_ERR_UNSUPPORTED = 5, /* valid input outside what the kernels are built for      */
    POLEE_ERR_COMM = 6         /* collective / communicator failure                       */
};

typedef struct polee_ctx polee_ctx;       /* device + stream + error state               */
typedef struct polee_ptt polee_ptt;       /* one Polya tree transform (device resident)  */
typedef struct polee_loglik polee_loglik; /* one sample's X, device resident             */
typedef struct polee_vi polee_vi;         /* state of one likelihood-approximation fit   */
typedef struct polee_approx polee_approx; /* S fitted approximations (regression input)  */
typedef struct polee_regression polee_regression; /* regression model + surrogate posterior */

/* ---- context -------------------------------------------------------------------- */
polee_status polee_ctx_create(int device, polee_ctx **out);
void polee_ctx_destroy(polee_ctx *ctx);
const char *polee_last_error(const polee_ctx *ctx_or_null);
polee_status polee_ctx_synchronize(polee_ctx *ctx);
/* free / total device memory of the context's GPU (hipMemGetInfo), in bytes; either pointer may be NULL */
polee_status polee_ctx_mem_info(polee_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);
void *polee_ctx_stream(polee_ctx *ctx); /* the context's hipStream_t */
/* HIP-event stopwatch on the context's stream (bench.py times the hot path with it). */
polee_status polee_ctx_timer_start(polee_ctx *ctx);
polee_status polee_ctx_timer_stop(polee_ctx *ctx, double *elapsed_ms);
const char *polee_version(void);
/* The host-side builders (polee_loglik_create, polee_hclust*) keep the large scratch blocks they used for the next sample
 * (mapping and unmapping gigabytes per sample was a third of a sample's preparation): up to POLEE_HOST_CACHE_MB
 * megabytes; default = a quarter of the memory available to the process (MemAvailable, cut to the cgroup's
 * memory.max - memory.current), at most 8192.  A long-lived process therefore keeps up to that much resident after a
 * sample's preparation.  polee_host_cache_trim releases the blocks; polee_host_cache_configure sets a new cap in MB
 * (freeing what no longer fits; < 0: only query) and returns the cap in force; polee_host_cache_bytes = bytes cached now. */
void polee_host_cache_trim(void);
int64_t polee_host_cache_configure(int64_t cap_mb);
int64_t polee_host_cache_bytes(void);
/* DEVICE buffers likewise: what the library frees on the device (the builders' scratch, handles' arrays) is kept by size class
 * for the next allocation instead of going through hipFree / hipMalloc (a device-wide wait each, and multi-second stalls every few
 * samples) -- up to POLEE_DEVICE_CACHE_MB megabytes (default 65536; 0: off).  polee_host_cache_trim frees these too;
 * polee_device_cache_bytes = device bytes kept now. */
int64_t polee_device_cache_bytes(void);

/* ---- Polya tree transform ---------------------------------------------------------
 * Replaces PolyaTreeTransform (src/ptt.jl:6-27) and the three TF custom ops of
 * src/tensorflow_ext/hsb_ops.cpp.  The tree is given exactly as serialised in the
 * prep HDF5 (src/ptt.jl:89-116): N = 2n-1 nodes in DFS pre-order, right child first;
 * node_parent_idxs[i] = 1-based parent (0 for the root), node_js[i] = 1-based
 * transcript id of a leaf (0 for an internal node).  The k-th internal node in node
 * order owns ys[k]. */
polee_status polee_ptt_create(polee_ctx *ctx, const int32_t *node_parent_idxs,
                              const int32_t *node_js, int32_t N, polee_ptt **out);
/* Same tree given as the TF-op index arrays (src/ptt.jl:293-309; hsb_ops.cpp:17-22). */
polee_status polee_ptt_create_from_index(polee_ctx *ctx, const int32_t *left_index,
                                         const int32_t *right_index, const int32_t *leaf_index,
                                         int32_t N, polee_ptt **out);
void polee_ptt_destroy(polee_ptt *t);
int32_t polee_ptt_n(const polee_ptt *t); /* number of leaves (transcripts) */
/* make_inverse_ptt_params (src/ptt.jl:293-309), host only. */
polee_status polee_make_inverse_ptt_params(const int32_t *node_parent_idxs, const int32_t *node_js,
                                           int32_t N, int32_t *left_index, int32_t *right_index,
                                           int32_t *leaf_index);

/* transform! (src/ptt.jl:125-160): ys f64 [B][n-1] in (0,1) -> xs f32 [B][n] on the
 * simplex, leaves floored at 1e-16; ladj (optional, [B]) = sum over internal nodes of
 * log u.  The node values u of the last call stay on the device for
 * polee_ptt_transform_gradients, like t.us does in the reference. */
polee_status polee_ptt_transform(polee_ptt *t, const double *ys, int32_t B, float *xs,
                                 double *ladj_or_null);
/* transform_gradients! (src/ptt.jl:167-209) when with_ladj != 0, else
 * transform_gradients_no_ladj! (src/ptt.jl:217-251).  x_grad f64 [B][n] ->
 * y_grad f64 [B][n-1] (the reference stores f32; callers may round). */
polee_status polee_ptt_transform_gradients(polee_ptt *t, const double *ys, const double *x_grad,
                                           int32_t B, int with_ladj, double *y_grad);
/* inverse_transform! (src/ptt.jl:257-285): xs f32 [B][n] -> ys f64 [B][n-1],
 * ladj [B] = -sum log u. */
polee_status polee_ptt_inverse_transform(polee_ptt *t, const float *xs, int32_t B, double *ys,
                                         double *ladj_or_null);
/* TF op HSB (hsb_ops.cpp:17-120): logits f32 [B][n-1] -> x f32 [B][n]; no floor. */
polee_status polee_hsb(polee_ptt *t, const float *y_logit, int32_t B, float *x);
/* TF op InvHSB (hsb_ops.cpp:128-249): x f32 [B][n] -> y f64 [B][n-1], ladj f32 [B]. */
polee_status polee_inv_hsb(polee_ptt *t, const float *x, int32_t B, double *y, float *ladj);
/* TF op InvHSBGrad (hsb_ops.cpp:252-402): VJP of InvHSB. */
polee_status polee_inv_hsb_grad(polee_ptt *t, const double *y_grad, const float *ladj_grad,
                                const double *y, int32_t B, float *backprops);

/* ---- sparse fragment x transcript log-likelihood -----------------------------------
 * Replaces Model + log_likelihood + factored_log_likelihood (src/likelihood.jl:2-85)
 * and pAt_mul_B!/pAt_mulinv_B! (src/sparse.jl:6-40).  X is m x n in CSC form exactly
 * as stored in the likelihood-matrix HDF5 (src/rnaseq_sample.jl:505-519): colptr
 * [n+1] and rowval [nnz] 1-based; colptr may be uint32 (as the reference's UInt32
 * index type) or uint64 (colptr_bytes = 4 or 8).  ks (optional, [m]) are the integer
 * row multiplicities of the factored likelihood. */
polee_status polee_loglik_create(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr,
                                 int colptr_bytes, const uint32_t *rowval, const float *nzval,
                                 const int64_t *ks_or_null, polee_loglik **out);
/* WHERE the device layout is built.  By default on the DEVICE (csrc/psell_device.hip): X is uploaded as given, transposed there
 * (a stable sort by fragment) and laid out by kernels that follow the host builder (csrc/psell_build.cpp) stage by stage -- byte
 * for byte the same layout (tests/test_gpu_device_build.py; 0.21 s at 20 M fragments x 200 k transcripts, against 0.73 s for the
 * host builder + upload).  The host builder takes over for a matrix without structure (a tenth of its non-zeros or more in
 * fragments that share no transcript set with their neighbours: stream C's question is sequential over all of them), for more
 * than 2^32 - 2 non-zeros, and when POLEE_DEVICE_BUILD=0 or one of its own experiment knobs (POLEE_PSELL_*) is set.
 * polee_loglik_built_on_device: 1 / 0 for a handle. */
int polee_loglik_built_on_device(const polee_loglik *ll);
/* Same sample given as Xt (n x m CSC == X in CSR), which the reference materialises
 * anyway (likelihood-approximation.jl:407): tcolptr [m+1] uint64 1-based row offsets,
 * trowval [nnz] 1-based transcript ids. */
polee_status polee_loglik_create_from_xt(polee_ctx *ctx, int64_t m, int64_t n,
                                         const uint64_t *tcolptr, const uint32_t *trowval,
                                         const float *tnzval, const int64_t *ks_or_null,
                                         polee_loglik **out);
void polee_loglik_destroy(polee_loglik *ll);

typedef struct {
    int64_t m, n, nnz;
    int64_t num_slices;      /* 64-row slices of the device layout                      */
    int64_t num_tiles;       /* workgroup-sized groups of slices                        */
    int64_t padded_nnz;      /* stored entries including padding                        */
    int64_t device_bytes;    /* bytes of X resident in HBM                              */
    int64_t stream_bytes;    /* bytes one likelihood pass streams from HBM (X only)     */
    int64_t num_empty_rows;  /* fragments with no compatible transcript (skipped)       */
    int32_t max_row_nnz;
    int32_t max_tile_cols;   /* largest per-tile column dictionary                      */
    /* the row streams of the device layout (see polee_amd/csrc/loglik_internal.hpp):                     */
    /* [0] dense uniform slices, sets of <= 16 transcripts; [1] masked uniform slices,   */
    /* unions of <= 16 (fragments whose sets differ); [2] dense uniform, 17..32; [3] masked uniform,    */
    /* unions of 17..32; [4] mixed slices of unrelated fragments of <= 15 transcripts -- [0..4] are one */
    /* persistent launch; [5] mixed slices of longer fragments (more than 32 transcripts, as a rule): a */
    /* second launch; [6] rows kept in CSR (fragments without any structure, whose sliced forms would   */
    /* cost more than CSR: a random sparse matrix): a launch of their own, no tiles; [7] fragments       */
    /* compatible with ONE transcript: collapsed at build time into a count per transcript (each adds    */
    /* log X_ij + log x_j to lp and 1 / x_j to the gradient): stream_bytes_hbm[7] = 4 n, no tiles        */
    int64_t stream_rows[8];  /* fragments                                               */
    int64_t stream_nnz[8];   /* non-zeros of X                                          */
    int64_t stream_tiles[8]; /* tiles (workgroup-sized units of work)                   */
    int64_t stream_bytes_hbm[8]; /* bytes of the slice stream a pass reads                */
    int64_t dict_entries;    /* entries of all tile dictionaries: x is gathered into, and the gradient flushed */
                             /* from, one window of dict_entries x K floats per pass                           */
} polee_loglik_info;
polee_status polee_loglik_get_info(const polee_loglik *ll, polee_loglik_info *info);
/* Deterministic mode ("fixed reduction order ... bitwise stable"): the gradient (and lp) of a pass is
 * summed in a fixed order -- per wave, per tile, then per transcript in tile order -- instead of with float atomics, so
 * that two evaluations on the same inputs agree bit for bit (and with them a whole fit with the same noise), whichever
 * workgroup takes which tile.  About 11 % slower (three workgroups per CU instead of four, a second small kernel).  Rows kept
 * in CSR (polee_loglik_info.stream_rows[6] > 0: fragments without any set structure, none in any input met so far) are
 * still added with float atomics: the guarantee holds when that stream is empty.  Default off. */
polee_status polee_loglik_set_deterministic(polee_loglik *ll, int on);

/* log_likelihood (src/likelihood.jl:36-56) for K expression vectors at once:
 * xs f32 [K][n] -> x_grad f64 [K][n] = sum_i X_ij / s_i  (x ks_i if factored);
 * lp [K] = sum_i log s_i (x ks_i) unless lp_or_null == NULL ("gradonly"). 1 <= K <= 8.
 * A polee_loglik supports ONE evaluation in flight at a time (polee_loglik_eval, or a polee_vi / polee_regression step that uses
 * the handle): the dynamic tile schedule keeps a per-handle counter on the device and its base on the host, and two threads or
 * streams evaluating the same handle at once would hand out wrong tiles.  Different handles are independent. */
polee_status polee_loglik_eval(polee_loglik *ll, const float *xs, int32_t K, double *x_grad,
                               double *lp_or_null);
/* effective_length_jacobian_adjustment! (src/likelihood.jl:93-110), batched:
 * x_grad[k][j] -= n / (efflens[j] * sum_i xs[k][i]/efflens[i]); xls optional out. */
polee_status polee_efflen_jacobian_adjustment(polee_ctx *ctx, const float *efflens, const float *xs,
                                              int32_t K, int64_t n, double *x_grad,
                                              float *xls_or_null);

/* gene_noninformative_prior! (src/likelihood.jl:114-159), batched over K draws: gradient of the
 * non-informative prior over gene-level expression.  gene_of int32[n] = gene index of each transcript
 * (0-based, -1 = no gene known): the reference's Dict{gene id -> transcript indexes} as an array.
 * xls f32 [K][n] is the output of polee_efflen_jacobian_adjustment; x_grad f64 [K][n] is updated. */
polee_status polee_gene_noninformative_prior(polee_ctx *ctx, const float *efflens, const float *xls,
                                             const float *xs, int32_t K, int64_t n, const int32_t *gene_of,
                                             double *x_grad);

/* ---- element-wise reparameterisations (standalone forms) --------------------------------
 * The Julia functions of src/logitnormal.jl:8-55, src/sinh_arcsinh.jl:10-38 and
 * src/kumaraswamy.jl:27-78 (inside the VI loop the first two are fused into the tree kernels).
 * Gradient entry points ACCUMULATE into *_grad like the reference's `+=`.  ladj pointers may be
 * NULL (= Val(false)). */
polee_status polee_logit_normal_transform(polee_ctx *ctx, const float *mu, const float *sigma,
                                          const float *zs, int64_t len, double *ys, double *ladj_or_null);
polee_status polee_logit_normal_transform_gradients(polee_ctx *ctx, const float *zs, const double *ys,
                                                    const float *sigma, const float *y_grad, int64_t len,
                                                    float *z_grad_or_null, float *mu_grad, float *sigma_grad);
polee_status polee_sinh_asinh_transform(polee_ctx *ctx, const float *alpha, const float *zs0, int64_t len,
                                        float *zs, double *ladj_or_null);
polee_status polee_sinh_asinh_transform_gradients(polee_ctx *ctx, const float *zs0, const float *alpha,
                                                  const float *z_grad, int64_t len, float *alpha_grad);
polee_status polee_kumaraswamy_transform(polee_ctx *ctx, const float *as, const float *bs, const float *zs,
                                         int64_t len, double *ys, double *ladj_or_null);
polee_status polee_kumaraswamy_transform_gradients(polee_ctx *ctx, const float *zs, const float *as,
                                                   const float *bs, const float *y_grad, int64_t len,
                                                   float *a_grad, float *b_grad);

/* ---- likelihood approximation (the VI loop) ----------------------------------------
 * Replaces approximate_likelihood(::LogitSkewNormalPTTApprox, sample)
 * (src/likelihood-approximation.jl:395-575), its factored variant (:248-392), ADAM
 * (:107-146) and the element-wise reparameterisations (src/logitnormal.jl:8-55,
 * src/sinh_arcsinh.jl:10-38).  Defaults are the reference's constants
 * (src/constants.jl:48-65, likelihood-approximation.jl:421-423). */
typedef struct {
    int32_t num_steps;           /* LIKAP_NUM_STEPS = 500                                  */
    int32_t num_mc_samples;      /* LIKAP_NUM_MC_SAMPLES = 6 (1..8)                        */
    int32_t use_efflen_jacobian; /* default 1 (--no-efflen-jacobian clears it)             */
    int32_t gradonly;            /* default 1: no ELBO / log-likelihood values             */
    uint64_t seed;               /* device Philox seed (reference default 123456789)       */
    const float *z0;             /* optional HOST noise [num_steps][num_mc][n-1]; when set */
                                 /* it replaces the device RNG (deterministic parity runs) */
    double y_eps;                /* LIKAP_Y_EPS = 1e-10 clamp of ys and xs; must lie in (0, 0.5) */
    double adam_initial_learning_rate; /* 1.0  */
    double adam_learning_rate_decay;   /* 2e-2 */
    double adam_min_learning_rate;     /* 1e-3 */
    double adam_eps;                   /* 1e-8 */
    double adam_rv;                    /* 0.9  */
    double adam_rm;                    /* 0.7  */
    double max_mu_step, max_omega_step, max_alpha_step; /* 0.2, 0.2, 0.02 */
    int32_t profile;             /* 1: bracket every sparse-kernel launch with HIP events; N > 1: every N-th launch (four
                                    event records per pass cost ~20 us of stream gaps per iteration at C2) */
    int32_t deterministic;       /* 1: this fit's likelihood passes run in deterministic mode (the handle's own  */
                                 /* polee_loglik_set_deterministic setting is untouched); -1: float atomics;      */
                                 /* 0 (default): deterministic exactly when the sample is shared by more than    */
                                 /* one rank (polee_vi_set_comm), so that repeated N-rank fits are bitwise equal  */
    const int32_t *gene_of;      /* optional HOST int32[n]: gene index of every transcript (0-based, -1 = none   */
                                 /* known) = gene_noninformative = true (likelihood-approximation.jl:475-491,     */
                                 /* 535-538: gene_noninformative_prior! after the effective-length adjustment,   */
                                 /* which it needs: use_efflen_jacobian must be on).  NULL: off (the CLI default) */
} polee_vi_opts;
void polee_vi_default_opts(polee_vi_opts *opts);

typedef struct {
    int32_t steps_done;
    int32_t nonfinite_step;        /* first step with a non-finite gradient, 0 if none    */
    double loglik_kernel_ms_avg;   /* profile=1: mean duration of the dominant sparse     */
                                   /* kernel (the persistent launch over the uniform streams) */
    int64_t loglik_kernel_launches;
    double last_elbo;              /* !gradonly: reference-style elbo of the last step    */
    double last_lp_mean;           /* !gradonly: mean log-likelihood over the K draws     */
    double loglik_pass_ms_avg;     /* profile=1: mean duration of one whole likelihood    */
                                   /* pass: the x-window gather, the persistent launch     */
                                   /* and, if the sample has mixed tiles, their launch     */
} polee_vi_stats;

/* Builds the state and the initial values mu = logit(inverse_transform(1/n)),
 * omega = log 0.1, alpha = 0 (likelihood-approximation.jl:451-456). */
polee_status polee_vi_create(polee_loglik *ll, polee_ptt *t, const float *efflens,
                             const polee_vi_opts *opts, polee_vi **out);
void polee_vi_destroy(polee_vi *vi);
/* Enqueues nsteps VI iterations (K draws + one ADAM update each) on the context's
 * stream without host synchronisation. */
polee_status polee_vi_run(polee_vi *vi, int32_t nsteps);
/* Waits for the stream; returns POLEE_ERR_NONFINITE if any step met a non-finite
 * gradient (likelihood-approximation.jl:559). */
polee_status polee_vi_sync(polee_vi *vi);
polee_status polee_vi_get_params(polee_vi *vi, float *mu, float *omega, float *alpha);
polee_status polee_vi_set_params(polee_vi *vi, const float *mu, const float *omega,
                                 const float *alpha);
polee_status polee_vi_get_stats(polee_vi *vi, polee_vi_stats *stats);
/* Per-step values recorded when gradonly == 0: elbo [steps_done] as the reference
 * computes it (last draw's lp + ladj, divided by K: likelihood-approximation.jl:537,
 * 561) and the mean log-likelihood over draws.  Either pointer may be NULL. */
polee_status polee_vi_get_trace(polee_vi *vi, double *elbo, double *lp_mean);
/* The noise the device RNG uses at (step, draw): z0 f32 [K][n-1] for 1-based step. */
polee_status polee_vi_export_noise(polee_vi *vi, int32_t step, float *z0);
/* Test hook: evaluates the K draws of the NEXT step at the current parameters and
 * returns the step's averaged gradients (what ADAM would consume), without updating
 * anything.  Any output may be NULL.  xs [K][n], x_grad [K][n] (after the effective
 * length adjustment), y_grad [K][n-1], *_grad [n-1], lp [K], ladj [K]. */
polee_status polee_vi_eval_gradients(polee_vi *vi, float *xs, double *x_grad, double *y_grad,
                                     float *mu_grad, float *omega_grad, float *alpha_grad,
                                     double *lp, double *ladj);
/* approximate_likelihood in one call: create + run(num_steps) + sync + get_params. */
polee_status polee_vi_fit(polee_loglik *ll, polee_ptt *t, const float *efflens,
                          const polee_vi_opts *opts, float *mu, float *omega, float *alpha,
                          polee_vi_stats *stats_or_null);

/* approximate_likelihood(::OptimizePTTApprox, sample) (src/likelihood-approximation.jl:149-242): point
 * optimisation of the expression vector by ADAM ascent on z (ys = logistic(z)), used by the reference to
 * assign reads while fitting bias models (src/rnaseq_sample.jl:343).  The reference builds a :sequential
 * tree itself; here the tree is the caller's.  xs f32 [n] (clamped to [1e-10, 1]); zs optional f32 [n-1]. */
polee_status polee_optimize_ptt(polee_loglik *ll, polee_ptt *t, const float *efflens, int32_t num_steps,
                                float *xs, float *zs_or_null);

/* ---- tree construction (host side) --------------------------------------------------------
 * hclust + order_nodes (src/hclust.jl:193-319, 361-389), the heuristic behind
 * PolyaTreeTransform(X, :cluster) (src/ptt.jl:35-52): greedy joining of the transcripts / subtrees that share
 * the most reads (Jaccard similarity of read sets among 25 neighbours in median-read order), remaining
 * components smallest first, nodes in DFS pre-order, right child first.  X in CSC, 1-based, as in the
 * likelihood-matrix HDF5.  Outputs int32 [2n-1] each: exactly the arrays polee_ptt_create takes and the prep
 * HDF5 stores.  Runs on the CPU (as the reference's does); no context needed. */
polee_status polee_hclust(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                          int32_t *node_parent_idxs, int32_t *node_js);
/* The same rule -- join the subtrees that share the most reads first -- in parallel rounds: every round merges each edge
 * that is the best edge (similarity, then a hash of the endpoints: a total order) of both its endpoints, on all host
 * threads; new nodes are numbered by their edges' priorities, so the tree does not depend on the thread count.  Not the
 * reference's tree node for node (its order depends on heap positions among equal similarities, which no parallel
 * schedule reproduces, and a locally best edge is merged before a better one can appear at an endpoint): a documented
 * variant for sample preparation at scale (polee_amd/csrc/hclust.cpp; the exact mode above stays the default).  Same
 * arguments, same output arrays. */
polee_status polee_hclust_parallel(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                   int32_t *node_parent_idxs, int32_t *node_js);
/* The same tree -- node for node the arrays polee_hclust_parallel gives -- built on the DEVICE (csrc/hclust_device.hip): per round the
 * best live edge of every node by atomic maxima over the priorities, the mutually-best edges sorted by priority, their read
 * sets united and the similarities to the neighbours of both halves counted by binary searches spread over the whole GPU. */
polee_status polee_hclust_parallel_device(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                          const uint32_t *rowval, int32_t *node_parent_idxs, int32_t *node_js);

/* ---- one sample over several GPUs -----------------------------------
 * X's rows (fragments) are sharded over the ranks in contiguous blocks; every rank creates its
 * polee_loglik from its block and runs the same polee_vi (same seed => identical state); per
 * likelihood pass the partial gradients (K*n f32) and log-likelihoods are summed with one
 * all-reduce over RCCL (bound at run time).  One rank calls polee_comm_unique_id and the caller
 * distributes the 128 bytes (MPI, torch.distributed, a file ...). */
typedef struct polee_comm polee_comm;
#define POLEE_COMM_ID_BYTES 128
polee_status polee_comm_unique_id(uint8_t id[POLEE_COMM_ID_BYTES]);
polee_status polee_comm_create(polee_ctx *ctx, int32_t nranks, int32_t rank,
                               const uint8_t id[POLEE_COMM_ID_BYTES], polee_comm **out);
/* The same communicator interface over a HOST all-reduce supplied by the caller (MPI_Allreduce, a torch.distributed
 * gloo group, ...): the library stages the buffer through host memory (stream-synchronising; for clusters without
 * RCCL between the ranks, and for tests that run two ranks on one GPU).  `allreduce(user, buf, count, is_f64)` must sum
 * `buf` (count f32 / f64 values) over all ranks in place and return 0. */
typedef int (*polee_host_allreduce_fn)(void *user, void *buf, int64_t count, int is_f64);
polee_status polee_comm_create_host(polee_ctx *ctx, int32_t nranks, int32_t rank, polee_host_allreduce_fn allreduce,
                                    void *user, polee_comm **out);
void polee_comm_destroy(polee_comm *comm);
int32_t polee_comm_rank(const polee_comm *comm);
int32_t polee_comm_size(const polee_comm *comm);
/* what the transport itself reports: *transport = 1 RCCL / 2 host-staged; *count, *user_rank = ncclCommCount,
 * ncclCommUserRank of the RCCL communicator (the creation arguments for a host communicator); any pointer may be NULL */
polee_status polee_comm_info(const polee_comm *comm, int32_t *transport, int32_t *count, int32_t *user_rank);
/* sum over ranks of a host buffer (convenience; the VI loop reduces device buffers in place) */
polee_status polee_allreduce_sum_f32(polee_comm *comm, float *buf, int64_t count);
/* make `vi` a row-sharded fit: its likelihood handle holds this rank's block of rows */
polee_status polee_vi_set_comm(polee_vi *vi, polee_comm *comm_or_null);

/* ---- sampler ------------------------------------------------------------------------
 * rand!(::ApproxLikelihoodSampler) (src/approx-sampler.jl:37-44): draws x f32
 * [ndraws][n] from a fitted approximation.  z0 (optional host [ndraws][n-1]) replaces
 * the device RNG.  No clamp, leaves floored at 1e-16 by transform!. */
polee_status polee_sampler_draw(polee_ptt *t, const float *mu, const float *sigma,
                                const float *alpha, const float *z0_or_null, int32_t ndraws,
                                uint64_t seed, float *xs);

/* Initial values of the model entry, load_samples_hdf5 (src/estimate.jl:436-455): the mean of ndraws (30 there) draws
 * z0 -> sinh-asinh -> logistic, y clamped to [LIKAP_Y_EPS, 1 - LIKAP_Y_EPS] -> transform!, each draw divided by the
 * effective lengths and renormalised; x0 f32 [n].  z0 as in polee_sampler_draw. */
polee_status polee_sampler_initial_values(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                          const float *efflens, const float *z0_or_null, int32_t ndraws,
                                          uint64_t seed, float *x0);

/* posterior_mean (src/approx-sampler.jl:86-117): mean over ndraws draws, each clamped to [1e-15, 0.9999999]
 * (f32 accumulation in draw order, as the reference); pm f32 [n].  z0 as in polee_sampler_draw. */
polee_status polee_sampler_posterior_mean(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                          const float *z0_or_null, int32_t ndraws, uint64_t seed, float *pm);
/* Statistics.quantile(loaded_samples, transforms, qs, N) for one sample (src/approx-sampler.jl:50-83): element-wise
 * quantiles (Julia's default definition, linear interpolation between order statistics) of ndraws draws;
 * qs f64 [nq] (nq <= 8), quantiles f32 [nq][n]. */
polee_status polee_sampler_quantiles(polee_ptt *t, const float *mu, const float *sigma, const float *alpha,
                                     const float *z0_or_null, int32_t ndraws, uint64_t seed, const double *qs,
                                     int32_t nq, float *quantiles);

/* ---- density of fitted approximations (regression consumer) --------------------------
 * Replaces RNASeqApproxLikelihoodDist._log_prob (src/polee_approx_likelihood.py:367-450)
 * and, with it, the InvHSB/InvHSBGrad ops inside TF's autodiff.  S samples, each with
 * its own tree (or one shared tree when shared_tree != 0: index arrays are [1][N]).
 * Arrays follow create_tensorflow_variables! (src/estimate.jl:502-556): efflens
 * [S][n], la_mu/la_sigma/la_alpha [S][n-1], left/right/leaf_index int32 [S][N]. */
polee_status polee_approx_create(polee_ctx *ctx, int32_t S, int32_t n, const float *efflens,
                                 const float *la_mu, const float *la_sigma, const float *la_alpha,
                                 const int32_t *left_index, const int32_t *right_index,
                                 const int32_t *leaf_index, int shared_tree, polee_approx **out);
void polee_approx_destroy(polee_approx *ap);
/* x f32 [S][n] unnormalised log-expression -> lp f32 [S]; x_grad (optional, [S][n]) =
 * d lp[s] / d x[s][:]. */
polee_status polee_approx_logprob(polee_approx *ap, const float *x, float *lp,
                                  float *x_grad_or_null);
/* Same on device buffers, enqueued on the context's stream (no host sync). */
polee_status polee_approx_logprob_device(polee_approx *ap, const float *d_x, float *d_lp,
                                         float *d_x_grad_or_null);
/* RNASeqGeneApproxLikelihoodDist (src/polee_gene_expression.py:14-90): the same density reached through
 * gene-level expression.  x[s][i] = x_gene[s][g(i)] + x_isoform[s][i] - logsumexp_{i' in g(i)} x_isoform[s][i']
 * (the reference's exp / blockwise sparse matmul / log), then _log_prob of the transcript approximation.
 * gene_of int32 [n]: 0-based gene of every transcript (each transcript in exactly one gene, every gene
 * non-empty).  lp f32 [S]; gradients (optional) d lp[s] / d x_gene [S][G] and / d x_isoform [S][n]. */
polee_status polee_approx_gene_logprob(polee_approx *ap, const float *x_gene, const float *x_isoform,
                                       const int32_t *gene_of, int32_t num_genes, float *lp,
                                       float *gene_grad_or_null, float *isoform_grad_or_null);
/* rnaseq_approx_likelihood_sampler (src/polee_approx_likelihood.py:35-59): one draw per
 * sample, divided by efflens, renormalised and clipped to [1e-16, 0.99999999].
 * z0 optional host [S][n-1]. */
polee_status polee_approx_sample(polee_approx *ap, const float *z0_or_null, uint64_t seed,
                                 float *x);

/* approximate_feature_likelihood (src/polee_gene_expression.py:191-222): a normal approximation of the log
 * expression of FEATURES (sets of transcripts, typically genes), by moments of sampler draws: loc = mean over
 * num_mean_draws draws of log sum_{t in feature} x_t (x from polee_approx_sample's sampler), scale = sqrt of the mean
 * squared deviation from loc over a further num_var_draws draws.  The feature / transcript incidence comes as
 * num_pairs (feature, transcript) pairs, 1-based (transcript_expression_to_feature_expression, :163-173).
 * z0 (optional, tests) = noise of all draws, host [(num_mean_draws + num_var_draws)][S][n-1].
 * loc, scale: f32 [S][F].  (The reference runs 1000 + 1000 draws.) */
polee_status polee_approx_feature_moments(polee_approx *ap, const int32_t *feature_idxs, const int32_t *transcript_idxs,
                                          int64_t num_pairs, int32_t F, int32_t num_mean_draws, int32_t num_var_draws,
                                          uint64_t seed, const float *z0_or_null, float *loc, float *scale);

/* approximate_splicing_likelihood (src/polee_splicing.py:47-113): the same moments for splicing log-ratios,
 * log sum_{t in feature} x_t - log sum_{t in antifeature} x_t.  feature_indices / antifeature_indices: int32 [P][2] /
 * [Q][2] rows (feature, transcript), 0-based, as the reference's NumPy arrays. */
polee_status polee_approx_splicing_moments(polee_approx *ap, const int32_t *feature_indices, int64_t num_feature_pairs,
                                           const int32_t *antifeature_indices, int64_t num_antifeature_pairs, int32_t F,
                                           int32_t num_mean_draws, int32_t num_var_draws, uint64_t seed,
                                           const float *z0_or_null, float *loc, float *scale);

/* ---- regression model ---------------------------------------------
 * Replaces RNASeqLinearRegression.__init__ / model_fn / variational_model_fn / fit
 * (models/polee_regression.py:18-340): the horseshoe+ linear model of log expression with kernel-regression
 * mean-variance and distortion terms, its mean-field surrogate posterior, and
 * tfp.vi.fit_surrogate_posterior(sample_size = 1, Adam(2e-3)) as hand-derived gradients of
 * loss = log q(z) - log p(z) at one reparameterised draw z per step.
 *   ap            fitted approximations of the S samples (the likelihood term, polee_approx_likelihood.py:367-450);
 *                 NULL or use_point_estimates != 0: no likelihood term, x fixed at x_init (qx_loc not trained)
 *   design        f32 [S][F] factor matrix;  x_init f32 [S][n] log expression;  sample_scales f32 [S]
 *   x_init_mean   f32 [n] column means of x_init over ALL samples, or NULL to take them from x_init (needed when
 *                 this handle holds one rank's shard of the samples, see polee_regression_set_comm)
 *   hinges        f32 [degree] kernel-regression knots, or NULL for choose_knots(min, max of the column means of
 *                 x_init) as RNASeqTranscriptLinearRegression does (models/polee_regression.py:436-440)
 *   x_bias_loc0 / x_bias_scale0   prior of x_bias (log(1/n) and 12 in the reference's subclasses)
 * Flat parameter (and gradient) vector, each array row-major, in this order (the reference's variable names
 * without "_var"):  qw_global_scale_variance_loc, .._softplus_scale, qw_global_scale_noncentered_loc,
 * .._softplus_scale (4 scalars); qw_distortion_c_loc [F][degree]; qx_scale_concentration_c_loc [degree];
 * qx_scale_scale_c_loc [degree]; then [F][n] each: qw_local1_scale_variance_{loc,softplus_scale},
 * qw_local1_scale_noncentered_{..}, qw_local2_scale_variance_{..}, qw_local2_scale_noncentered_{..}, qw_loc,
 * qw_softplus_scale; then [n] each: qx_bias_loc, qx_bias_softplus_scale, qx_scale_loc, qx_scale_softplus_scale;
 * then [S][n] each: qx_loc, qx_softplus_scale.
 * Noise vector (standard normals of the draw): w_global_scale_variance, w_global_scale_noncentered (2 scalars);
 * [F][n] each: w_local1_scale_variance, w_local1_scale_noncentered, w_local2_.., w_local2_.., w; x_bias [n];
 * x_scale [n]; x [S][n]. */
polee_status polee_regression_create(polee_ctx *ctx, polee_approx *ap_or_null, int32_t S, int32_t F, int32_t n,
                                     const float *design, const float *x_init, const float *x_init_mean_or_null,
                                     const float *sample_scales, const float *hinges_or_null, int32_t degree,
                                     float bandwidth, float x_bias_loc0, float x_bias_scale0, int use_distortion,
                                     float scale_penalty, int use_point_estimates, polee_regression **out);
void polee_regression_destroy(polee_regression *reg);
int64_t polee_regression_num_params(const polee_regression *reg);
int64_t polee_regression_num_noise(const polee_regression *reg);
polee_status polee_regression_get_params(polee_regression *reg, float *params);
polee_status polee_regression_set_params(polee_regression *reg, const float *params);
/* RNASeqNormalTranscriptLinearRegression (models/polee_regression.py:490-531): replace the likelihood term by
 * point estimates with a scale, loc[s][j] ~ Normal(log softmax(x[s])[j], scale[s][j]); loc, scale f32 [S][n]. */
polee_status polee_regression_set_normal_likelihood(polee_regression *reg, const float *loc, const float *scale);
/* RNASeqGeneLinearRegression (models/polee_regression.py:533-600): the model's n features are GENES and the
 * likelihood is RNASeqGeneApproxLikelihoodDist (polee_approx_gene_logprob) of (x_gene, x_isoform), with
 * x_isoform_mean ~ Normal(0, 2) [nt], x_isoform ~ Normal(x_isoform_mean, 1) [S][nt] and Normal surrogates for both.
 * Create the model with ap = NULL over the genes, then attach: ap over the nt transcripts, gene_of int32 [nt]
 * (0-based gene of every transcript), x_isoform_init f32 [S][nt].  Adds an isoform block of parameters (own Adam
 * state): qx_isoform_mean_loc [nt], qx_isoform_mean_softplus_scale [nt], qx_isoform_loc [S][nt],
 * qx_isoform_softplus_scale [S][nt]; and appends noise (mean [nt], isoform [S][nt]) to the noise vector. */
polee_status polee_regression_set_gene_likelihood(polee_regression *reg, polee_approx *ap, const int32_t *gene_of,
                                                  const float *x_isoform_init);
/* RNASeqGeneIsoformLinearRegression (models/polee_regression.py:656-877): as above, but the isoform block is a
 * regression of its own over the nt transcripts: x_isoform ~ Normal(x_isoform_bias + F_isoform w_isoform,
 * x_isoform_scale) with horseshoe+ coefficients w_isoform [Fi][nt] (:698-720), x_isoform_bias ~ Normal(0, 2) (:722-724),
 * x_isoform_scale ~ InverseGamma(0.001, 0.001) (:728-729) and the surrogates of :736-833.  design_isoform f32 [S][Fi].
 * The isoform block then holds, in this order (the model's own order with the isoform design and no hinges):
 * global scale variance loc / softplus_scale, global scale noncentered loc / softplus_scale; [Fi][nt] arrays local1
 * variance loc / s, local1 noncentered loc / s, local2 variance loc / s, local2 noncentered loc / s, w loc / s;
 * [nt] arrays bias loc / s, scale loc / s; [S][nt] arrays x_isoform loc / s.  Noise appended to the noise vector: 2
 * globals, five [Fi][nt] arrays (the four scales, w), bias [nt], scale [nt], x_isoform [S][nt]. */
polee_status polee_regression_set_gene_isoform_likelihood(polee_regression *reg, polee_approx *ap,
                                                          const int32_t *gene_of, const float *x_isoform_init,
                                                          const float *design_isoform, int32_t num_isoform_factors);
/* RNASeqJointLinearRegression (models/polee_regression.py:879-1283; driven by models/joint-regression.jl): gene-level
 * regression + a regression over SPLICE FEATURES whose predictor reaches the transcripts through the 0/1 feature matrix.
 * Create the model over the gene features with ap = NULL, use_distortion = 0, scale_penalty = 5e-4 (:1052-1054), then
 * attach: ap over the nt transcripts, gene_of int32 [nt], x_isoform_init f32 [S][nt], the number of splice features P and
 * the matrix's non-zeros as (transcript, feature) pairs, 0-based.  The gene block becomes the joint model's: a horseshoe
 * prior (ONE local scale level: the local2 arrays of the flat vector stay, unused), kernel-regression weights that follow
 * the SAMPLED bias (:1034-1035), HalfCauchy(0, 10) on the mean-variance coefficients (:1037-1041),
 * qw_softplus_scale = -2, Adam(1e-3) (:1215).  The isoform block then holds: the splice block in the model's own order
 * with P columns and no hinges -- global scale variance loc / s, noncentered loc / s; [F][P] arrays local variance loc / s,
 * local noncentered loc / s, (local2: four unused arrays), w_splice loc / s; [P] arrays bias loc / s, (x_scale: two
 * unused arrays) -- then x_iso_scale loc / s [nt] (SoftplusNormal; prior HalfCauchy(0, 1), :1110-1112) and x_iso loc / s
 * [S][nt] (prior Normal(x_iso_loc, x_iso_scale), :1114-1116).  Noise appended to the noise vector: 2 globals, five [F][P]
 * arrays, bias [P], (unused [P]), x_iso_scale [nt], x_iso [S][nt].  One GPU only. */
polee_status polee_regression_set_joint_likelihood(polee_regression *reg, polee_approx *ap, const int32_t *gene_of,
                                                   const float *x_isoform_init, int32_t num_splice_features,
                                                   const int32_t *pair_transcript, const int32_t *pair_feature,
                                                   int64_t num_pairs);
/* Adam's learning rate of polee_regression_fit (default 2e-3: models/polee_regression.py:326; the joint model sets 1e-3) */
polee_status polee_regression_set_learning_rate(polee_regression *reg, float learning_rate);
int64_t polee_regression_num_isoform_params(const polee_regression *reg);
polee_status polee_regression_get_isoform_params(polee_regression *reg, float *params);
polee_status polee_regression_set_isoform_params(polee_regression *reg, const float *params);
/* gradient of the isoform block left by the last polee_regression_eval */
polee_status polee_regression_get_isoform_grad(polee_regression *reg, float *grad);
/* Samples sharded over ranks: every rank creates the model over ITS samples (S = local count, the
 * same F, n, hinges, x_init_mean and seed everywhere), so the shared parameters are replicas and qx_* are local.
 * Per step one sum all-reduce of (F+2) n + 32 f32 observation-model statistics is the only exchange. */
polee_status polee_regression_set_comm(polee_regression *reg, polee_comm *comm_or_null);
/* kernel_regression_weights (src/polee.py:36-47) as the model uses them: f32 [degree][n] */
polee_status polee_regression_weights(polee_regression *reg, float *weights);
/* loss and (optionally) its gradient w.r.t. the flat parameter vector at the current parameters, for the draw
 * defined by `noise` (host, num_noise values) or, when NULL, by the device RNG with `seed`; no update. */
polee_status polee_regression_eval(polee_regression *reg, const float *noise_or_null, uint64_t seed, float *loss,
                                   float *grad_or_null);
/* classify (models/polee_regression.py:342-413): the design matrix of the (testing) samples is a latent variable -- a relaxed one-hot
 * row per sample whose logits are the only new trainable quantity -- while everything the fitted model shares stays fixed.  The
 * host side (polee_amd/regression.py, RNASeqLinearRegression.classify; the Julia caller models/classify.jl does the same through
 * PyCall) draws the relaxed rows and owns the logits; the device model over the testing samples provides:
 *   _set_design     the step's design matrix, f32 [S][F] (replaces the one given at creation; from then on every evaluation also
 *                   computes d loss / d design -- the observation model's term, -sum_j a[s][j] w_eff[f][j]);
 *   _set_trainable  Adam of polee_regression_fit moves the flat parameters [begin, end) only (the testing samples' qx_loc /
 *                   qx_softplus_scale blocks; begin == end with point estimates); default: all of them;
 *   _design_grad    d loss / d design of the last evaluation or fit step, f32 [S][F].
 * Transcript-level model on one GPU (the gene-level classify, :601-651, is not built). */
polee_status polee_regression_set_design(polee_regression *reg, const float *design);
polee_status polee_regression_set_trainable(polee_regression *reg, int64_t begin, int64_t end);
polee_status polee_regression_design_grad(polee_regression *reg, float *grad);
/* niter steps of fit() (models/polee_regression.py:303-340): draw, loss + gradient, Adam(2e-3, 0.9, 0.999, 1e-7).
 * noise (optional, tests): host [niter][num_noise].  loss_trace (optional): f32 [niter]. */
polee_status polee_regression_fit(polee_regression *reg, int32_t niter, uint64_t seed, const float *noise_or_null,
                                  float *loss_trace_or_null);

/* ---- construction of the likelihood matrix (first slice) ------------------------------------
 * Replaces, for pre-parsed inputs, the reference's intersection of alignment pairs with transcripts and the
 * conditional fragment probabilities of its SimplisticFragModel (bias terms = 1):
 *   parallel_intersection_loop (src/rnaseq_sample.jl:58-121), fragmentlength (src/transcripts.jl:273-446),
 *   effective_length / condfragprob (src/fragmodel.jl:119-169), sortperm + compact_indexes! + sparse
 *   (src/rnaseq_sample.jl:126-157, 470-489).
 * BAM / GFF parsing, bias models and read assignment stay upstream.  Coordinates are 1-based inclusive.
 *   transcripts  j = 0..n-1 (= t.metadata.id - 1): sequence id, strand (+1 / -1), exons ascending and disjoint
 *   fragments    i = 0..m-1 = alignment pairs: sequence id, strand, the LEFTMOST mate (m1) and, for paired-end reads,
 *                the other one (m2; m2_left == 0: single-end); per mate its CIGAR as (operation code, length) pairs
 *                in BAM coding (0 M, 1 I, 2 D, 3 N, 4 S), cig?_ptr[i] .. cig?_ptr[i+1] into cig_op / cig_len (an
 *                empty range = one match over [left, right]); m1_is_flag16: the lone mate's SAM flag is exactly 16
 *                (the reference's test `aln.flag == SAM.FLAG_REVERSE != 0`, src/fragmodel.jl:130)
 *   fragmodel    fraglen_pmf / fraglen_cdf f32 [2000] (index l-1 holds length l), the median, strand specificity,
 *                alt_frag_model (src/fragmodel.jl:23-115)
 * Result: X's rows compressed (what polee_loglik_create_from_xt takes): tcolptr u64 [rows+1] 1-based, trowval u32
 * 1-based transcript ids ascending within a row, tnzval f32; effective_lengths f32 [n]; row_fragment i64 [rows] =
 * the fragment of every row (fragments compatible with no transcript are dropped, as compact_indexes! does; rows keep
 * the fragments' input order -- the reference numbers them in the order of its interval trees, a permutation the
 * likelihood does not see). */
typedef struct {
    int32_t n;
    const int32_t *seq;
    const int8_t *strand;
    const int64_t *exon_ptr;   /* [n+1] */
    const int64_t *exon_first, *exon_last;
} polee_xb_transcripts;
typedef struct {
    int64_t m;
    const int32_t *seq;
    const int8_t *strand;
    const int64_t *m1_left, *m1_right, *m2_left, *m2_right;
    const uint8_t *m1_is_flag16;
    const int64_t *cig1_ptr, *cig2_ptr;  /* [m+1] each */
    const uint8_t *cig_op;
    const int32_t *cig_len;
} polee_xb_fragments;
typedef struct {
    const float *fraglen_pmf, *fraglen_cdf;
    int32_t fraglen_median;
    float strand_specificity;
    int32_t alt_frag_model;
} polee_xb_fragmodel;
/* A TRAINED bias model for the reference's default BiasedFragModel (src/fragmodel.jl:174-445; training, src/bias.jl:266-786,
 * stays upstream).  tseq: the transcripts' spliced sequences in transcript orientation (t.metadata.seq), one code per base:
 * 0 A, 1 C, 2 G, 3 T, 4 anything else; tseq_ptr [n+1] (tseq_ptr[0] = 0, lengths = exonic lengths).  Sequence bias
 * (SeqBiasModel{:left} / {:right}, bias.jl:157-160, 419-456): seqbias_len = 20 positions, orders [20] (-1: position not in
 * the model), ps [20][4][ps_ctx] row-major (the reference's ps[i, c+1, ctx+1]), ps_ctx >= 4^(largest order).  gc_bins: the
 * SimpleHistogramModel's bins (bias.jl:459-521).  pos_terms [pos_maxtlen] + pos_p: PositionalBiasModel (bias.jl:523-659) or
 * NULL (use_pos_bias = false, the default).  high_prob_fraglens: the BIAS_EFFLEN_NUM_FRAGLENS most probable fragment lengths
 * in descending probability (fragmodel.jl:358-359).  m1_reverse [m]: the lone mate of a single-end fragment has
 * FLAG_REVERSE set (transcripts.jl:486); may be NULL when every fragment is paired.
 * Not reproducible: context positions beyond a transcript's ends are RANDOM nucleotides in the reference (bias.jl:83-85,
 * 424-429); here they read as A. */
typedef struct {
    const int64_t *tseq_ptr;
    const uint8_t *tseq;
    int32_t seqbias_len, ps_ctx;
    const int32_t *orders_left, *orders_right;
    const float *ps_left, *ps_right;
    int32_t gc_nbins;
    const float *gc_bins;
    double pos_p;
    const double *pos_terms;
    int32_t pos_maxtlen;
    int32_t num_fraglens;
    const int32_t *high_prob_fraglens;
    const uint8_t *m1_reverse;
} polee_xb_biasmodel;
typedef struct polee_xbuild polee_xbuild;
polee_status polee_xbuild_run(polee_ctx *ctx, const polee_xb_transcripts *transcripts, const polee_xb_fragments *fragments,
                              const polee_xb_fragmodel *fragmodel, polee_xbuild **out);
/* The same under the BiasedFragModel: compute_transcript_bias! (bias.jl:834-858), effective_length (fragmodel.jl:372-410),
 * condfragprob with genomic_to_transcriptomic (fragmodel.jl:413-445, transcripts.jl:452-538).  polee_xbuild_get_bias
 * returns the transcripts' bias vectors (left / right f32 [tseq_ptr[n]], laid out like tseq) and the kernel's time. */
polee_status polee_xbuild_run_biased(polee_ctx *ctx, const polee_xb_transcripts *transcripts, const polee_xb_fragments *fragments,
                                     const polee_xb_fragmodel *fragmodel, const polee_xb_biasmodel *biasmodel, polee_xbuild **out);
polee_status polee_xbuild_get_bias(const polee_xbuild *xb, float *left_bias_or_null, float *right_bias_or_null, double *ms_bias_or_null);
void polee_xbuild_destroy(polee_xbuild *xb);
/* rows and non-zeros of the result; kernel times (ms): effective lengths, counting pass, filling pass */
polee_status polee_xbuild_sizes(const polee_xbuild *xb, int64_t *rows, int64_t *nnz, double *ms_efflen, double *ms_count,
                                double *ms_fill);
/* copies the result to host arrays (any pointer may be NULL) */
polee_status polee_xbuild_get(const polee_xbuild *xb, uint64_t *tcolptr, uint32_t *trowval, float *tnzval,
                              float *effective_lengths, int64_t *row_fragment);
/* X (by columns, 1-based: polee_loglik_create's arguments) uploaded ONCE for the two device builders that read it when a sample is
 * prepared from host arrays -- the tree (PolyaTreeTransform(X, :cluster), ptt.jl:35-52 -> hclust.jl:180-330) and the layout
 * (RNASeqSample, likelihood.jl:2-41's X).  polee_loglik_create + polee_hclust_parallel_device each upload their own copy: 2.9 GB over
 * PCIe at C2 instead of 1.9 GB.  The handle may be used from two contexts of the same device at once (the builders only read it) and
 * must outlive both calls.  POLEE_ERR_UNSUPPORTED when rows or non-zeros need more than 32 bits (the host paths take those). */
typedef struct polee_devx polee_devx;
polee_status polee_devx_upload(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                               const float *nzval, polee_devx **out);
/* nzval may be NULL in polee_devx_upload and follow here: the tree needs colptr + rowval only and can start while the values are on
 * their way (what sample_and_tree of the Python mirror does) */
polee_status polee_devx_upload_values(polee_devx *dx, const float *nzval);
void polee_devx_destroy(polee_devx *dx);
/* as polee_loglik_create / polee_hclust_parallel_device on the arrays the handle was made from: the same layout byte for byte, the same tree.
 * nzval_or_null: the values, if the handle has none yet -- they go up beside the layout's first kernels (row keys, sort, column search
 * need colptr + rowval only) and stay in the handle: at most ONE call that passes values may run on a handle at a time (the
 * handle's value buffer is written by it; calls on a handle that already holds its values only read it and may run side by side).
 * With the device layout builder switched off (POLEE_DEVICE_BUILD=0 or a host-builder knob) the layout is built on the host from
 * the handle's arrays, as polee_loglik_create would. */
polee_status polee_loglik_create_from_devx(polee_ctx *ctx, polee_devx *dx, const float *nzval_or_null, const int64_t *ks_or_null, polee_loglik **out);
polee_status polee_hclust_parallel_device_from_devx(polee_ctx *ctx, const polee_devx *dx, int32_t *node_parent_idxs, int32_t *node_js);
/* The likelihood handle straight from an xbuild result, without X leaving the device: rows_to_device -> layout kernels
 * (as polee_loglik_create_from_xt on polee_xbuild_get's arrays; ks_or_null: host array [rows]).  The xbuild handle stays valid. */
polee_status polee_loglik_create_from_xbuild(polee_ctx *ctx, const polee_xbuild *xb, const int64_t *ks_or_null, polee_loglik **out);
/* ... and the tree (polee_hclust_parallel_device) from the same result on the device: alignments -> X -> tree + layout -> fit
 * without X visiting the host (the columns of X by a stable sort of the result's rows by transcript). */
polee_status polee_hclust_parallel_device_from_xbuild(polee_ctx *ctx, const polee_xbuild *xb, int32_t *node_parent_idxs, int32_t *node_js);

#ifdef __cplusplus
}
#endif
#endif /* POLEE_HIP_H */
