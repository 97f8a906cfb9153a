// Sparse fragment x transcript likelihood: device layout ("PSELL") and internal entry points.
//
// PSELL = pattern-sorted sliced ELL with tile-local column dictionaries.
//   * Fragments (rows of X) are reordered: key = (first transcript / 256, row length,
//     hash of the row's transcript set).  Reordering rows changes no result (the
//     likelihood is a sum over fragments); it puts fragments with the same compatible
//     transcript set next to each other and keeps rows of equal length together.
//   * 64 consecutive rows form a SLICE, one row per lane of a wavefront, stored
//     column-major:  float val[w][64]; uint16 lcol[w][64];   w = longest row of the slice
//     (shorter rows are padded with val = 0).  One wave-instruction therefore loads 256
//     contiguous bytes of values and 128 of indices -- fully coalesced, no row pointers.
//   * Consecutive slices form a TILE (64 / 64 / 32 / 16 / 64 / 16 slices in streams A1 / A1M / A2 / A2M / BN / B), the unit of work of one
//     256-thread workgroup.  A tile owns a dictionary of the distinct transcripts its rows touch
//     (dict: local id u16 -> transcript id u32; the tile is closed before the dictionary would
//     pass 128 entries).  Every tile's dictionary starts at a multiple of 4 entries (padded with
//     transcript 0), so that its x window -- x[dict][K], gathered once per pass into one contiguous buffer
//     by xwin_gather_kernel -- starts 16-byte aligned and reaches LDS by LDS-DMA like the slice stream.
//   * The kernel accumulates the tile's gradient contributions in LDS (ds_add_f32) and flushes L*K
//     values to HBM per tile.
//   * Rows are split into six streams, each a contiguous range of tiles (in this order):
//       A1, A1M, A2, A2M = UNIFORM slices: all rows of a slice are stored under ONE transcript set, kept once per slice.
//         A1  (dense, sets of <= 16): uint16 lcol[128] header + float val[w][64], 4 B per entry: whole 64-row slices of
//              every run of identical rows, run remainders of >= 32 rows (zero padded), and dense UNION slices -- leftover
//              rows of neighbouring sets packed under the union of their sets with zeros where a row lacks a transcript
//              (a zero adds nothing to a row sum or a gradient, exactly) -- when the union is (almost) full.
//         A1M (MASKED, unions of <= 16): leftover rows whose sets differ -- any 64 rows whose union has <= 16 transcripts.
//              Header: uint32 hw[64] -- low half of hw[r] = the mask of fragment r (bit t: it is compatible with
//              transcript t of the union), high half of hw[t], t < 16, = the tile-local id of transcript t (0x8000 past
//              the union); then float val[i][64], the i-th non-zero of fragment r at position r of row i, for i < the
//              longest row of the slice.  (Every 32-bit word of a slice is a FINITE float -- ids < 128 or 0x8000 in the
//              high halves: the dense streams' kernels read the rows behind a slice's last transcript from their LDS
//              rings and multiply them by zero, and a ring may still hold a masked tile's bytes.)  Zeros cost no bytes: 4 B per non-zero + 4 B per
//              fragment + padding to the slice's longest row -- never more than the dense union slice, and below CSR's
//              8 B per non-zero + 4 B per fragment whatever the sets look like.  The kernel expands a fragment's values
//              with its mask (rank = popcount of the lower bits) and runs the same matrix-core phases.
//         A2  (dense, sets of 17..32): runs, remainders and union slices that are (almost) full.
//         A2M (MASKED, unions of 17..32): the leftover rows that fit no union of 16 -- fragments of a gene with many
//              isoforms.  Header: two rows of uint32 hw[64] (mask bits 0..15 / 16..31 in the low halves, the ids of
//              transcripts 0..15 / 16..31 in the high halves of each row's first sixteen words); values as in A1M.
//       BN, B = MIXED slices (float val[w][64]; uint16 lcol[w][64], 6 B per entry, any 64 fragments of the tile): the
//              leftover rows that fit no uniform slice at CSR's cost or less -- unrelated fragments, fewer than a handful
//              of which share any 32 transcripts.  BN: rows of <= 15 transcripts (with multiplicities a row float ks[64]
//              follows, at the next multiple of 256 bytes), lane = fragment, gathers from / LDS adds into the tile's
//              windows.
//         One persistent launch (loglik_stream_kernel) streams these five: the usual sample needs no other.
//              B: rows of more than 16 transcripts that found no uniform slice (more than 32, as a rule); the per-tile
//              kernel loglik_psell_kernel takes them in a second launch.  That kernel also runs over all other slices
//              on request (polee_debug_loglik_force_mixed): the independent second algorithm of the cross-check tests.
//       C     = rows kept in CSR (u32 ids, f32 values, u32 row offsets: 8 B per non-zero + 4 B per row): fragments with no structure at all,
//              whose mixed tiles would close on the dictionary before a slice is full -- the last resort that keeps
//              the layout below CSR's size for ANY matrix; loglik_csr_kernel (lane = row, global gathers / atomics).
//       S     = fragments compatible with ONE transcript (54 % of the rows of the reference's real-data fixture): not stored at
//              all.  Such a fragment adds log(X_ij) + log(x_j) to the log-likelihood and 1 / x_j to transcript j's
//              gradient (X_ij cancels: X_ij / (X_ij x_j)), ks_i times with multiplicities -- so the build keeps c_j = the
//              number of such fragments per transcript (single_cnt) and the constant sum of log X_ij (single_logsum), and
//              a pass adds c_j / x_j[k] and c_j log x_j[k] + const (single_rows_kernel): 4 B per TRANSCRIPT instead of
//              8 B + a slice lane per fragment.
//   * Each slice carries two flag bits (in the top bits of its offset word): "uniform" (its rows
//     are stored under one transcript set) and "continues" (the same set as the previous slice).  Runs of such
//     slices -- the bulk of real and synthetic data, where many fragments fall into the same
//     equivalence class -- keep their gradient contributions in registers.
// HBM traffic per likelihood pass ~ 4.9 B/nnz at BASELINE's C2 (CSR: 8 B + row pointers), read once.
#pragma once
#include "common.hpp"

namespace polee {

constexpr int PSELL_LANES = 64;
constexpr int PSELL_MAX_TILE_COLS = 1024;     // hard limit = longest supported row
constexpr int PSELL_TILE_COLS_TARGET = 128;   // a tile is closed when its dictionary would grow past this
constexpr int PSELL_DICT_ALIGN = 4;           // a tile's dictionary starts at a multiple of this many entries
constexpr int PSELL_TILE_SLICES_A1 = 64;      // slices per tile (= per workgroup) in each stream
constexpr int PSELL_TILE_SLICES_A2 = 32;     // (round 5: all four waves work on a wide tile -- 8 slices until then, for two)
constexpr int PSELL_TILE_SLICES_B = 16;
constexpr int PSELL_MAX_K = 8;
constexpr uint32_t PSELL_OFF_MASK = 0x1fffffffu;  // slice_off entries carry the slice flags in bits 29..31
constexpr uint32_t PSELL_FLAG_MASKED_BIT = 29;    // bit 29: a MASKED slice (set for every slice of streams A1M / A2M; the cross-check kernel reads it)
constexpr int PSELL_NARROW_MAX = 16;        // widest transcript set of stream A1 (7 KiB LDS ring, four groups of four transcripts: narrow_stream)
constexpr int PSELL_WIDE_MAX = 32;          // widest transcript set of streams A2 / A2M (two 16-row MFMA tiles: wide_stream, in two stages through the 7 KiB ring; A2M: 14 KiB rings)
// Uniform slices store fragment r of transcript row t at this position of the row's 64 values, chosen per stream so that
// the kernel's LDS operand reads are bank-conflict free: A1 (batched 4 x 4 outer products, narrow_stream) r ^ (t & 3);
// A2 (16 x 16 x 4 tiles, wide_stream) a rotation by 4 t.
enum : int { PSELL_A1 = 0, PSELL_A1M = 1, PSELL_A2 = 2, PSELL_A2M = 3, PSELL_BN = 4, PSELL_B = 5, PSELL_C = 6, PSELL_S = 7, PSELL_NSTREAMS = 8 };  // streams, in tile order (C and S have no tiles)
constexpr int PSELL_MIXED_NARROW_MAX = 15;  // longest row of stream BN (its slices pass through the narrow streams' 7 KiB rings)
constexpr int PSELL_TILE_SLICES_A2M = 16;
constexpr int PSELL_TILE_SLICES_BN = 64;
constexpr uint32_t psell_row_pos(int stream, uint32_t t, uint32_t r) { return stream == PSELL_A1 ? (r ^ (t & 3u)) : (stream == PSELL_A2 ? ((r + 4u * t) & 63u) : r); }
constexpr uint16_t PSELL_NO_COL = 0x8000u;  // header entries of a masked slice past its union (0x8000xxxx is a finite float)
// leftover rows are packed in independent chunks of this many candidates (host: a thread each; device: a WAVE each, whose walk is
// bound by instruction latency -- a group is cut once per chunk.  2 048 since round 6 (4 096 before): polee_loglik_create 0.080 / 0.096 s
// -> 0.073 / 0.084 s at C2 patterns / literal with the pass's time unchanged within run-to-run noise on the three inputs of
// profiles/r05_pack_chunk_ab.txt; 1 024 saves another 6 % of the build and costs the tiled real fixture 1 % of its pass)
constexpr uint32_t PSELL_PACK_CHUNK = 1u << 11;
// Stage 3 cuts the streams into segments whose tiles are made independently (a tile never spans two).  The mixed streams are walked
// ROW BY ROW (every row its own set), the uniform ones stretch by stretch: a mixed segment is this many rows, not the 2^18 of the
// others -- on the device a segment is one wave's work, and one 19 k-row segment of stream BN was the whole tile walk's time.
constexpr uint32_t PSELL_MIXED_SEG_ROWS = 1u << 12;
constexpr double PSELL_PACK_WIDE_RESERVE = 32768.0;  // bytes above CSR a part of the second pass may spend on rows too long for stream BN, beyond its allowance
constexpr int PSELL_MIN_UNIFORM_ROWS = 32;  // smallest run remainder stored as a padded uniform slice
constexpr int PSELL_MIN_UNION_ROWS = 1;     // smallest group of leftover rows stored as a union slice (1: every row of <= 32 transcripts is in a uniform slice)

// One position of the streaming kernel's static schedule (workgroup b walks positions b, b + grid, b + 2 grid, ...).
struct PosDesc {
    uint32_t tile;    // POS_NONE: the column ends here
    uint32_t s0, s1;  // slices of the tile
    uint32_t d0, L;   // first dictionary entry, entries in use
    uint32_t c1, c2, c3;  // uniform tiles: the waves stream slices [s0, c1), [c1, c2), [c2, c3), [c3, s1) (balanced on cost)
};
constexpr uint32_t POS_NONE = 0xffffffffu;

struct PsellHost {
    int64_t m = 0, n = 0, nnz = 0;
    int64_t num_slices = 0, num_tiles = 0, padded_nnz = 0, empty_rows = 0;
    // stream A (tiles [0, num_tiles_a)): only slices whose 64 rows share one transcript set
    int64_t rows_a = 0, num_tiles_a = 0, num_slices_a = 0;
    int64_t rows_a1 = 0, num_tiles_a1 = 0;    // A1 = tiles [0, num_tiles_a1): dense, sets of <= PSELL_NARROW_MAX transcripts
    int64_t rows_a1m = 0, num_tiles_a1m = 0;  // A1M = tiles [num_tiles_a1, num_tiles_a1m): masked
    int64_t rows_a2 = 0, num_tiles_a2 = 0;    // A2 = [num_tiles_a1m, num_tiles_a2): dense wide; A2M = [num_tiles_a2, num_tiles_a): masked wide
    int64_t rows_s = 0, num_tiles_s = 0;      // BN = [num_tiles_a, num_tiles_s): mixed narrow -- [0, num_tiles_s) is the persistent launch's share; B = the rest
    int64_t stream_rows[PSELL_NSTREAMS] = {}, stream_nnz[PSELL_NSTREAMS] = {}, stream_bytes[PSELL_NSTREAMS] = {};
    int stream_of_tile(int64_t t) const
    {
        return t < num_tiles_a1 ? PSELL_A1 : (t < num_tiles_a1m ? PSELL_A1M : (t < num_tiles_a2 ? PSELL_A2 : (t < num_tiles_a ? PSELL_A2M : (t < num_tiles_s ? PSELL_BN : PSELL_B))));
    }
    int32_t max_row = 0, max_tile_cols = 0;
    std::vector<uint8_t, default_init_allocator<uint8_t>> data;  // slice blocks (resize(n) leaves new bytes uninitialised; resize(n, 0) zeroes)
    std::vector<uint32_t> slice_off;   // [num_slices+1], 128-byte units in bits 0..29, slice flags in bits 30..31
    std::vector<uint32_t> tile_slice;  // [num_tiles+1]
    std::vector<uint32_t> tile_dict;   // [num_tiles+1], multiples of PSELL_DICT_ALIGN
    std::vector<uint32_t> tile_cols;   // [num_tiles] dictionary entries in use (the rest, up to tile_dict[t+1], is padding)
    std::vector<uint32_t> dict;        // transcript ids (0-based)
    std::vector<uint32_t> big_tiles;   // tiles whose dictionary exceeds PSELL_TILE_COLS_TARGET (a fragment with > 256 transcripts)
    std::vector<uint8_t> slice_flags;  // [num_slices] bit0 uniform, bit1 continues the previous slice's set, bit2 masked
    std::vector<uint8_t> slice_w;      // [num_slices] transcripts of the slice's set (uniform streams) / longest row (mixed)
    std::vector<float> slice_ks;       // optional [num_slices*64] row multiplicities
    std::vector<uint32_t> row_order;   // [stored rows] original 0-based row id per (slice, lane); ~0u = empty lane
    // stream C: rows kept in CSR (0-based transcript ids) because every sliced form would cost more (psell_build.cpp)
    std::vector<uint32_t> csr_rowptr;
    std::vector<uint32_t> csr_col, csr_rows;  // csr_rows: original row ids
    std::vector<float> csr_val, csr_ks;
    // stream S: fragments with exactly one compatible transcript, collapsed at build time
    std::vector<float> single_cnt;  // [n] (empty: none) sum of the multiplicities of transcript j's single-transcript fragments
    double single_logsum = 0.0;     // sum of ks_i log X_ij over them
    std::vector<uint32_t> single_rows;  // their original row ids, ascending (debug view / tests)
};

// The builder's stages (psell_build.cpp; psell_device.hip builds the same on the device and is checked against these):
struct PsellRuns {  // stage 1 -> 2: rows of the exact runs (with their slice ends) and the leftover rows, in sort-key order
    BVec<uint32_t> a1_rows, a1_ends, a2_rows, a2_ends, rb;
};
struct PsellRows {  // stage 2 -> 3: the ordered rows of the sliced streams (bounds: PsellHost::rows_a1 .. rows_s)
    BVec<uint32_t> rows, run_end, row_gid;  // run_end: 1 = the row's slice ends after it; row_gid: its group (forms 1, 2)
    BVec<uint8_t> row_form;                 // 0 exact run, 1 dense union, 2 masked
    BVec<uint32_t> pat_ptr, pat_col;        // the groups' transcript sets (unions)
};
int psell_bin_shift();
std::string psell_stage1(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                         const int64_t *ks, PsellHost &out, PsellRuns &R);
std::string psell_stage2(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                         const int64_t *ks, PsellRuns &R, PsellHost &out, PsellRows &W);
std::string psell_stage3(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                         const int64_t *ks, const PsellRows &W, PsellHost &out);
// Builds the layout from X in CSR form (0-based): rowptr [m+1], col [nnz], val [nnz].
// Returns "" or an error message.
std::string build_psell(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                        const int64_t *ks, PsellHost &out);

}  // namespace polee

namespace polee {
// A caller that numbers the transcripts differently from X's columns (the VI loop numbers them in the tree's leaf order, so
// that its x / gradient vectors are contiguous in the tree kernels) hands the pass the tables that name transcripts,
// translated into its numbering; x and g are then indexed by that numbering.
struct LoglikRemap {
    const uint32_t *dict;       // [dict_len] the tile dictionaries
    const float *single_cnt;    // [n] stream S's counts, or null
    const uint32_t *csr_col;    // [csr_nnz] stream C's column ids, or null
    const uint32_t *index_of;   // [n] transcript -> the caller's index (deterministic mode's per-transcript reduction)
    bool singles_in_g;          // the caller has already put cnt / x into g (the VI loop's forward kernel does)
};
}  // namespace polee

struct polee_xbuild;
namespace polee {
// xbuild.hip: the result of an xbuild run where it lies (Xt, 1-based, device memory)
polee_status xbuild_device_view(const polee_xbuild *xb, polee_ctx **ctx, int64_t *rows, int64_t *n, const uint64_t **tcolptr,
                                const uint32_t **trowval, const float **tnzval);
}  // namespace polee

// X by columns in device memory, uploaded ONCE for the two builders that read it (polee_devx_upload): the tree wants colptr + rowval,
// the layout all three.  1-based as the caller's arrays are; colptr widened to 64 bits.
struct polee_devx {
    polee_ctx *ctx = nullptr;
    int64_t m = 0, n = 0;
    uint64_t nnz = 0;
    polee::DevBuf<uint64_t> cp;
    polee::DevBuf<uint32_t> rowval;
    polee::DevBuf<float> nzval;
};

struct polee_loglik {
    polee_ctx *ctx = nullptr;
    int refs = 1;
    int64_t m = 0, n = 0, nnz = 0;
    bool has_ks = false;
    bool device_built = false;  // the slice stream was laid out on the device (psell_device.hip)
    bool force_mixed = false;  // debug: process every slice with the mixed-slice kernel
    bool xwin_ready = false;   // (per call) the x windows are already filled
    // (per call) a small job of the caller's that rides along with the x-window gather as one extra workgroup: out[d] = sum
    // over nparts rows of part[.][d], out[K + d] = its reciprocal (the VI loop's sum x / efflen).  side_done tells the caller
    // whether a launch took it.
    const double *side_part = nullptr;
    int side_nparts = 0;
    double *side_out = nullptr;
    bool side_done = false;
    polee::PsellHost host;  // metadata kept; bulk vectors are released after upload unless debugging
    polee::DevBuf<uint8_t> d_data;
    polee::DevBuf<uint32_t> d_slice_off, d_tile_slice, d_tile_dict, d_dict;
    polee::DevBuf<float> d_slice_ks;
    polee::DevBuf<uint32_t> d_csr_rowptr;  // stream C (rows kept in CSR)
    polee::DevBuf<uint32_t> d_csr_col;
    polee::DevBuf<float> d_csr_val, d_csr_ks;
    int64_t csr_rows = 0, csr_nnz = 0;
    polee::DevBuf<float> d_single_cnt;    // stream S: [n] multiplicity-weighted count of single-transcript fragments
    polee::DevBuf<double> d_single_part;  // per-block partial sums of c_j log x_j[k] (lp only)
    bool has_singles = false;
    const polee::LoglikRemap *cur_remap = nullptr;  // (per call) the caller's transcript numbering, see loglik_eval_device
    // the streaming kernel: per-pass x windows, static schedule (built for the grid of the first launch)
    polee::DevBuf<float> d_xwin;
    polee::DevBuf<polee::PosDesc> d_sched;
    int sched_grid = 0;
    polee::DevBuf<polee::PosDesc> d_sched_dyn;  // dynamic schedule: the tiles by descending cost + POS_NONE padding
    polee::DevBuf<unsigned int> d_dyn_ctr;      // [0]: positions drawn so far, running on from launch to launch ([1] unused)
    size_t dyn_pad = 0;
    size_t dyn_positions = 0;  // positions of the dynamic list = draws a launch makes
    uint32_t dyn_base = 0;     // the counter's value when the next launch starts (modulo 2^32)
    std::vector<float> tile_cost;  // relative cost of every tile (bytes it streams, weighted by stream)
    std::vector<uint32_t> tile_cut;  // [3 * num_tiles] slice boundaries between the waves of a uniform tile
    int64_t dict_len = 0;
    int occ_cache[polee::PSELL_MAX_K + 1][2][2][2] = {};
    // deterministic mode (polee_loglik_set_deterministic): per-tile gradient windows, per-workgroup lp sums, and the
    // dictionary entries of every transcript in tile order
    bool deterministic = false;
    polee::DevBuf<float> d_gwin;
    polee::DevBuf<double> d_lpwin;
    polee::DevBuf<uint32_t> d_tslot_ptr, d_tslot, d_theavy;  // (d_theavy: transcripts present in many tiles)
    // staging for the host-pointer API
    polee::DevBuf<float> d_x_rows, d_x_aos, d_g_aos;
    polee::DevBuf<double> d_g_rows, d_lp;
    // profiling of the sparse kernel
    bool profile = false;
    int prof_every = 1;        // bracket every prof_every-th pass with events (polee_vi_opts.profile = N)
    uint64_t prof_tick = 0;
    hipEvent_t cur_e0 = nullptr, cur_e1 = nullptr;  // bracket the dominant launch of the current pass
    hipEvent_t cur_p0 = nullptr, cur_p1 = nullptr;  // bracket the whole pass
    std::vector<hipEvent_t> prof_events;
    size_t prof_used = 0;
    int64_t prof_launches = 0;
    double prof_ms_total = 0.0, prof_pass_ms_total = 0.0;
    polee_status profile_collect();
};

namespace polee {
// d_x, d_g: [n][K] f32 (transcript-major, the K draws of one transcript adjacent).
// d_g must be zeroed by the caller; the kernel adds into it.  d_lp [K] f64 or null
// (also accumulated into).
void loglik_retain(polee_loglik *ll);
void loglik_release(polee_loglik *ll);
// xwin_ready: the caller has already written the tiles' x windows (ll->d_xwin, through the slot lists d_tslot_ptr / d_tslot:
// the VI loop's forward kernel does, saving the gather launch)
polee_status loglik_eval_device(polee_loglik *ll, const float *d_x, int K, float *d_g, double *d_lp, bool xwin_ready = false,
                                const LoglikRemap *remap = nullptr);
}  // namespace polee
