/*
 * polee_hip_debug.h -- host-only introspection of the device layouts libpolee_hip builds.
 * These entry points never touch the GPU; the CPU test-suite uses them to check the tree
 * plan (Euler tour, leaf ranges) and the PSELL matrix layout against the oracle by
 * emulating the kernels in NumPy.  Not part of the drop-in boundary.
 */
#ifndef POLEE_HIP_DEBUG_H
#define POLEE_HIP_DEBUG_H

#include "polee_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tree plan of polee_ptt_create (see polee_amd/csrc/ptt_internal.hpp for the encoding).
 * Caller-provided arrays: tour_code u32[3n-2], tour_tgt i32[3n-2], leaf_tid i32[n],
 * lo/mid/hi1 i32[n-1]; max_depth optional. */
polee_status polee_debug_ptt_plan(const int32_t *node_parent_idxs, const int32_t *node_js, int32_t N,
                                  uint32_t *tour_code, int32_t *tour_tgt, int32_t *leaf_tid,
                                  int32_t *lo, int32_t *mid, int32_t *hi1, int32_t *max_depth);

typedef struct polee_psell_debug polee_psell_debug;
typedef struct {
    int64_t m, n, nnz, num_slices, num_tiles, padded_nnz, num_empty_rows;
    int64_t data_bytes, dict_len;
    int32_t max_row_nnz, max_tile_cols;
    const uint8_t *data;        /* mixed slice blocks: float val[w][64]; uint16 lcol[w][64] */
    const uint32_t *slice_off;  /* [num_slices+1], 128-byte units                          */
    const uint32_t *tile_slice; /* [num_tiles+1]                                           */
    const uint32_t *tile_dict;  /* [num_tiles+1]                                           */
    const uint32_t *dict;       /* [dict_len] 0-based transcript ids                       */
    const uint32_t *row_order;  /* [num_slices*64] 0-based fragment per lane, ~0 = empty   */
    const float *slice_ks;      /* [num_slices*64] or NULL                                 */
    const uint8_t *slice_flags; /* [num_slices] bit0 uniform, bit1 continues previous      */
    int64_t num_tiles_a;        /* tiles [0, num_tiles_a) hold COMPACT slices:             */
                                /*   uint16 lcol[128] (w used); float val[w][64]           */
    int64_t num_tiles_a1;       /* tiles [0, num_tiles_a1): dense, transcript sets of <= 16 */
    int64_t num_tiles_a1m;      /* tiles [num_tiles_a1, num_tiles_a1m): MASKED slices:      */
                                /*   uint32 hw[64]: low half = mask of the lane's fragment, */
                                /*   high half (lanes < 16) = id of that transcript of the  */
                                /*   union, 0x8000 past it; float val[i][64] = i-th non-zero */
                                /*   of the lane's fragment                                 */
    int64_t num_tiles_a2;       /* [num_tiles_a1m, num_tiles_a2): dense, 17..32;            */
                                /* [num_tiles_a2, num_tiles_a): MASKED, unions of 17..32:   */
                                /*   two header rows hw[2][64] (mask bits 0..15 / 16..31,   */
                                /*   ids of transcripts 0..15 / 16..31)                     */
    const uint8_t *slice_w;     /* [num_slices] transcripts of the slice's set / longest row (mixed) */
    int64_t num_tiles_s;        /* [num_tiles_a, num_tiles_s): mixed slices of fragments of <= 15 transcripts (with
                                   multiplicities a row float ks[64] follows at the next multiple of 256 bytes):
                                   tiles [0, num_tiles_s) are the persistent launch's share                      */
    int64_t stream_rows[8], stream_nnz[8], stream_bytes[8]; /* as in polee_loglik_info */
    int64_t csr_num_rows;        /* stream C: rows kept in CSR (every sliced form would cost more) */
    const uint32_t *csr_rowptr;  /* [csr_num_rows + 1] 0-based                                     */
    const uint32_t *csr_col;     /* 0-based transcript ids                                         */
    const float *csr_val;
    const uint32_t *csr_rows;    /* [csr_num_rows] original 0-based fragment of every row          */
    int64_t single_num_rows;     /* stream S: fragments with ONE compatible transcript, collapsed at build time    */
    const uint32_t *single_rows; /* [single_num_rows] their original 0-based fragment ids, ascending               */
    const float *single_cnt;     /* [n] sum of their multiplicities per transcript, or NULL (none)                 */
    double single_logsum;        /* sum of ks_i log X_ij over them                                                 */
} polee_psell_view;
/* Same arguments as polee_loglik_create, minus the context. */
polee_status polee_debug_psell_build(int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                     const uint32_t *rowval, const float *nzval,
                                     const int64_t *ks_or_null, polee_psell_debug **out);
/* The same layout with some of the builder's stages run on the DEVICE (csrc/psell_device.hip; bit 0: keys / sort / runs,
 * bit 1: packing of leftover rows, bit 2: slices and tiles) and the others on the host, each continuing from the other's
 * output.  Every mix gives the bytes polee_debug_psell_build gives (tests/test_gpu_device_build.py). */
polee_status polee_debug_psell_build_device(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                            const uint32_t *rowval, const float *nzval, const int64_t *ks_or_null,
                                            int device_stages, polee_psell_debug **out);
polee_status polee_debug_psell_view(const polee_psell_debug *p, polee_psell_view *view);
void polee_debug_psell_free(polee_psell_debug *p);

/* Debug mode of the kept device buffers (POLEE_DEVICE_CACHE_POISON=1 in the environment before the library's first allocation;
 * csrc/common.hpp DevBlockCache): a released block is filled with a pattern on its owner's stream and the pattern is verified
 * when the block is handed out again or freed.  Counts so far: blocks verified, blocks found overwritten, words overwritten. */
void polee_debug_device_cache_poison(int64_t *checked_blocks, int64_t *bad_blocks, int64_t *bad_words);

/* Process every slice with the mixed-slice kernel (the uniform streams' LDS-DMA kernel is bypassed): two
 * different algorithms over the same layout, used by the full-size cross-check test. */
polee_status polee_debug_loglik_force_mixed(polee_loglik *ll, int on);

/* The two halves of polee_regression_eval with the exchange left to the caller (tests emulate sample sharding on one
 * GPU): data pass -> stats f32 [num_stats] ((F+2) n sums + a few loss slots) of this handle's samples; prior pass <- stats summed over the shards. */
int64_t polee_debug_regression_num_stats(const polee_regression *reg);
polee_status polee_debug_regression_data_pass(polee_regression *reg, const float *noise, float *stats);
polee_status polee_debug_regression_prior_pass(polee_regression *reg, const float *stats, float *loss,
                                               float *grad_or_null);

/* `reps` all-reduces of `count` f32 on the communicator's stream between two HIP events: *ms_avg per call (the first,
 * untimed call sets the connections up).  bench.py reports it next to the N > 1 rates. */
polee_status polee_debug_comm_allreduce_ms(polee_comm *comm, int64_t count, int32_t reps, double *ms_avg);

/* fast_log (csrc/scan.hpp), the double-precision log of the tree kernels, element-wise (tests check it against libm) */
polee_status polee_debug_fast_log(polee_ctx *ctx, const double *x, int64_t count, double *out);
/* fast_exp (csrc/scan.hpp), the double-precision exp of the VI loop's forward kernel (leaf u = exp of a path sum of edge logs) */
polee_status polee_debug_fast_exp(polee_ctx *ctx, const double *x, int64_t count, double *out);

#ifdef __cplusplus
}
#endif
#endif
