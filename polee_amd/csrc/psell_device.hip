// Device-side construction of the PSELL layout (loglik_internal.hpp): the stages of psell_build.cpp as HIP kernels, byte for
// byte the same layout -- the host builder is the checker (polee_debug_psell_build_device runs any mix of host and device
// stages and hands the result to the same debug view; tests/test_gpu_device_build.py).
// Plays the role of `Xt = SparseMatrixCSC(transpose(X))` in the reference (src/likelihood-approximation.jl:406-408).
//
// What is sequential in the host builder stays sequential here, one WAVE per independent piece (a segment of a stream in
// stage 3), with the 64 lanes spread over the transcripts of a set; everything per row, per slice and per byte is parallel.
#include <cstring>
#include <type_traits>
#include <chrono>
#include <cstdio>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "loglik_internal.hpp"
#include "psell_device.hpp"

namespace polee {
namespace {

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ inline uint64_t lanes_below() { return (1ull << lane_id()) - 1ull; }
template <typename T>
__device__ inline T bcast0(T v) { return __shfl(v, 0); }

struct SegDesc {
    uint32_t ra, rb, stream, pad;
};
struct SegAux {
    uint32_t st0, st1;  // stretches (maximal runs of rows under one transcript set) of the segment
    uint32_t sl0;       // uniform streams: index of its first slice in the numbering of the slice ends
    uint32_t tbase;     // scratch: tiles
    uint64_t dbase;     // scratch: dictionary entries
};
struct SegOut {
    uint32_t ntiles, nslices, ndict, pad;
};

struct S3In {
    const uint64_t *rowptr;
    const uint32_t *col;
    const float *val;
    const int64_t *ks;
    const uint32_t *rows, *run_end, *gid;
    const uint8_t *form;
    const uint32_t *pat_ptr, *pat_col;
    uint32_t Nr, n;
    uint32_t bounds[7];  // rows of stream st: [bounds[st], bounds[st + 1])
};

__device__ inline int stream_of_row(const S3In &A, uint32_t ri)
{
    int st = 0;
    while (st < 5 && ri >= A.bounds[st + 1]) ++st;
    return st;
}

// per ordered row: its length; head = it starts a new STRETCH (a segment or stream starts, or its transcript set -- its own, or
// its group's union -- differs from the previous row's); endflag = a slice of a uniform stream ends after it
__global__ void s3_rowinfo_kernel(S3In A, uint32_t *rlen, uint32_t *head, uint32_t *endflag)
{
    const uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x;
    if (ri >= A.Nr) return;
    const uint32_t r = A.rows[ri];
    const uint64_t b = A.rowptr[r];
    const uint32_t len = (uint32_t)(A.rowptr[r + 1] - b);
    rlen[ri] = len;
    const int st = stream_of_row(A, ri);
    const bool uniform = st <= PSELL_A2M;
    endflag[ri] = uniform && A.run_end[ri] != 0 ? 1u : 0u;
    uint32_t h = 1;
    if (uniform && ri > A.bounds[st]) {
        const bool g = A.form[ri] != 0, gp = A.form[ri - 1] != 0;
        if (g == gp) {
            if (g) {
                h = A.gid[ri] != A.gid[ri - 1];
            } else {
                const uint32_t rp = A.rows[ri - 1];
                const uint64_t bp = A.rowptr[rp];
                bool same = (uint32_t)(A.rowptr[rp + 1] - bp) == len;
                for (uint32_t k = 0; same && k < len; ++k) same = A.col[b + k] == A.col[bp + k];
                h = same ? 0u : 1u;
            }
        }
    }
    head[ri] = h;
}

// the streams cut into segments (psell_build.cpp, stage 3: "SEGMENTS of about a million rows", same cut points)
__global__ void s3_segments_kernel(S3In A, uint32_t seg_rows, SegDesc *segs, uint32_t *nseg, uint32_t *head, uint32_t max_segs)
{
    if (threadIdx.x || blockIdx.x) return;
    uint32_t k = 0;
    for (int st = 0; st < 6; ++st) {
        uint64_t a = A.bounds[st];
        const uint64_t end = A.bounds[st + 1];
        const uint64_t step = (st == PSELL_B || st == PSELL_BN) ? (seg_rows < PSELL_MIXED_SEG_ROWS ? seg_rows : PSELL_MIXED_SEG_ROWS) : seg_rows;
        while (a < end) {
            uint64_t e = a + step < end ? a + step : end;
            if (e < end) {
                if (st != PSELL_B && st != PSELL_BN) {
                    while (e < end && !A.run_end[e - 1]) ++e;
                } else {
                    const uint64_t tile_rows = (uint64_t)PSELL_LANES * PSELL_TILE_SLICES_B;
                    const uint64_t e2 = a + ((e - a + tile_rows - 1) / tile_rows) * tile_rows;
                    e = e2 < end ? e2 : end;
                }
            }
            if (k < max_segs) segs[k] = SegDesc{(uint32_t)a, (uint32_t)e, (uint32_t)st, 0u};
            head[a] = 1;
            ++k;
            a = e;
        }
    }
    *nseg = k;
}

__global__ void s3_scatter_kernel(uint32_t Nr, const uint32_t *head, const uint32_t *hscan, const uint32_t *endflag,
                                  const uint32_t *escan, uint32_t *st_start, uint32_t *endpos)
{
    const uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x;
    if (ri > Nr) return;
    if (ri == Nr) {
        st_start[hscan[Nr]] = Nr;
        return;
    }
    if (head[ri]) st_start[hscan[ri]] = ri;
    if (endflag[ri]) endpos[escan[ri]] = ri;
}

// per segment: its stretches, its first slice, and room in the scratch arrays (tiles: at most one per slice / per row;
// dictionary entries: at most 32 + padding per (stretch, tile) pair in the uniform streams, the rows' own entries + padding in
// the mixed ones)
__global__ void s3_segaux_kernel(S3In A, const SegDesc *segs, uint32_t nseg, const uint32_t *hscan, const uint32_t *escan,
                                 const uint64_t *lenps, SegAux *aux, uint64_t *totals)
{
    if (threadIdx.x || blockIdx.x) return;
    uint64_t tb = 0, db = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        const SegDesc s = segs[k];
        SegAux x;
        x.st0 = hscan[s.ra];
        x.st1 = hscan[s.rb];
        x.sl0 = escan[s.ra];
        x.tbase = (uint32_t)tb;
        x.dbase = db;
        aux[k] = x;
        if (s.stream <= PSELL_A2M) {
            const uint64_t nsl = escan[s.rb] - escan[s.ra];
            tb += nsl;
            db += (uint64_t)(PSELL_WIDE_MAX + PSELL_DICT_ALIGN) * (nsl + (x.st1 - x.st0));
        } else {
            tb += s.rb - s.ra;
            db += (lenps[s.rb] - lenps[s.ra]) + (uint64_t)PSELL_DICT_ALIGN * (s.rb - s.ra);
        }
    }
    totals[0] = tb;
    totals[1] = db;
}

struct S3Seq {
    S3In A;
    const SegDesc *segs;
    const SegAux *aux;
    uint32_t nseg;
    const uint32_t *rlen, *st_start, *escan;
    uint32_t *stamps;  // [waves][n], uniform streams: the tile in which a transcript was last registered (the tile ids of a wave never repeat)
    uint32_t *t_s0, *t_cols, *t_dstart;
    uint32_t *dict_s;
    uint32_t *srec_ri, *srec_n;  // mixed streams: first row and number of rows of every slice, at (ra - bounds[BN]) + slice
    SegOut *outs;
    uint32_t *next_seg;
    uint32_t caps[6];
};

// TILES AND DICTIONARIES: the one sequential part of stage 3 (psell_build.cpp, emit_segment: a row joins the current tile
// while the tile's dictionary holds its transcripts or has room for them; a tile closes on its slice count).  One wave per
// segment.  In the uniform streams the slices are known beforehand (they end where run_end says), and all rows of a stretch
// ask the same question of the dictionary -- so the wave steps from stretch to stretch, not from row to row.
// (Round 5, the MIXED streams -- row by row, every row its own set.  "Is this transcript in the current tile's dictionary?" was a table
// of stamps in global memory and the walk a chain of dependent global loads, four per row (row id, row start, column, stamp -- the
// stamp twice), 1.1 us per row: ONE segment of 11 - 19 k rows of stream BN took 12 - 21 ms on one wave while the ~115 others, 3 - 5 ms
// each, had long finished.  Now, for these streams: the dictionary is a hash set in LDS (the same question, the same answers, the same
// output); the rows' starts, lengths and -- for rows of at most 16 columns, all of stream BN -- columns arrive 64 rows at a time, a lane
// each, all loads of a batch in flight together, handed over through LDS; the short rows' loop body holds no global load, so nothing
// waits for the dictionary STORES (a vmcnt(0) per row was ~1 us); and a mixed segment is 4 096 rows (PSELL_MIXED_SEG_ROWS).  Tile
// walk at C2: 12 / 21.5 ms -> 5.3 ms, the uniform streams' longest segment.  The same batching of the uniform streams' stretches was
// tried and hung the GPU on the first test input for a reason not found in the time there was: not kept.)
constexpr uint32_t S3_BCOLS = 16;
constexpr uint32_t S3_TAB_BITS = 12, S3_TAB = 1u << S3_TAB_BITS;  // (a tile's dictionary holds at most PSELL_MAX_TILE_COLS = 1 024 entries)
static_assert(PSELL_MAX_TILE_COLS * 2 <= (int)S3_TAB, "the hash set of a tile's dictionary");

__global__ __launch_bounds__(64) void s3_tiles_kernel(S3Seq P)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[S3_TAB];  // mixed streams: column + 1 of the current tile's dictionary entries, 0 = free
    __shared__ uint32_t scol[64][PSELL_WIDE_MAX + 1];  // the column sets of a batch of 64 stretches (uniform streams) or 64 short rows (mixed)
    const uint32_t lane = lane_id();
    uint32_t *stamp = P.stamps + (size_t)blockIdx.x * P.A.n;  // uniform streams: the tile in which a transcript was last registered
    uint32_t tile_id = 1;
    auto tab_clear = [&]() {
        for (uint32_t i = lane; i < S3_TAB; i += 64) tab[i] = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto tab_has = [&](uint32_t c) -> bool {
        uint32_t slot = (c * 2654435761u) >> (32 - S3_TAB_BITS);
        for (uint32_t probes = 0; probes < S3_TAB; ++probes) {  // (bounded: a full table cannot occur, and must not hang the GPU if it does)
            const uint32_t v = tab[slot];
            if (v == c + 1u) return true;
            if (v == 0u) return false;
            slot = (slot + 1u) & (S3_TAB - 1u);
        }
        return false;
    };
    auto tab_put = [&](uint32_t c) {  // (c is not in the set; other lanes insert other columns at the same time)
        uint32_t slot = (c * 2654435761u) >> (32 - S3_TAB_BITS);
        for (uint32_t probes = 0; probes < S3_TAB && atomicCAS(&tab[slot], 0u, c + 1u) != 0u; ++probes) slot = (slot + 1u) & (S3_TAB - 1u);
    };
    for (;;) {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(P.next_seg, 1u);
        k = bcast0(k);
        if (k >= P.nseg) break;
        const SegDesc seg = P.segs[k];
        const SegAux aux = P.aux[k];
        const int st = (int)seg.stream;
        const uint32_t cap = P.caps[st];
        const unsigned long long t_seg0 = wall_clock64();  // (POLEE_BUILD_TIMING: the segment's time in SegOut.pad, 10 ns units)
        uint32_t ntile = 0, nsl = 0, dict_n = 0, tile_cols = 0, pending = 0, tile_d0 = 0, tile_s0 = 0;
        uint32_t *dict_out = P.dict_s + aux.dbase;
        auto close_tile = [&]() {
            if (!pending) return;
            if (lane == 0) {
                P.t_s0[aux.tbase + ntile] = tile_s0;
                P.t_cols[aux.tbase + ntile] = tile_cols;
                P.t_dstart[aux.tbase + ntile] = tile_d0;
            }
            ++ntile;
            dict_n = (dict_n + PSELL_DICT_ALIGN - 1) & ~(uint32_t)(PSELL_DICT_ALIGN - 1);
            tile_d0 = dict_n;
            tile_s0 = nsl;
            ++tile_id;
            tile_cols = 0;
            pending = 0;
            if (st > PSELL_A2M) tab_clear();
        };
        if (st <= PSELL_A2M) {
            for (uint32_t s = aux.st0; s < aux.st1; ++s) {
                const uint32_t a = P.st_start[s], b = P.st_start[s + 1];
                uint32_t j = P.escan[a];
                const uint32_t j1 = P.escan[b];
                const uint32_t *set;
                uint32_t w;
                if (P.A.form[a]) {
                    const uint32_t g = P.A.gid[a];
                    set = P.A.pat_col + P.A.pat_ptr[g];
                    w = P.A.pat_ptr[g + 1] - P.A.pat_ptr[g];
                } else {
                    set = P.A.col + P.A.rowptr[P.A.rows[a]];
                    w = P.rlen[a];
                }
                const uint32_t c = lane < w ? set[lane] : 0u;  // (w <= 32 in the uniform streams)
                while (j < j1) {
                    const bool fr = lane < w && stamp[c] != tile_id;
                    const uint64_t bal = __ballot(fr);
                    const uint32_t fresh = (uint32_t)__popcll(bal);
                    if (!(tile_cols + fresh <= (uint32_t)PSELL_TILE_COLS_TARGET || (tile_cols == 0 && pending == 0))) {
                        close_tile();
                        continue;
                    }
                    if (fr) {
                        stamp[c] = tile_id;
                        dict_out[dict_n + (uint32_t)__popcll(bal & lanes_below())] = c;
                    }
                    dict_n += fresh;
                    tile_cols += fresh;
                    const uint32_t take = min(j1 - j, cap - pending);
                    pending += take;
                    nsl += take;
                    j += take;
                    if (pending >= cap) close_tile();
                    __threadfence_block();
                }
            }
            close_tile();
        } else {
            const uint32_t sbase = seg.ra - P.A.bounds[PSELL_BN];
            uint32_t in_slice = 0, slice_start = seg.ra;
            auto close_slice = [&]() {
                if (!in_slice) return;
                if (lane == 0) {
                    P.srec_ri[sbase + nsl] = slice_start;
                    P.srec_n[sbase + nsl] = in_slice;
                }
                ++nsl;
                ++pending;
                in_slice = 0;
            };
            tab_clear();
            for (uint32_t ri0 = seg.ra; ri0 < seg.rb; ri0 += 64) {
                const uint32_t nb = min(64u, seg.rb - ri0);
                unsigned long long my_b = 0ull;
                uint32_t my_w = 0u;
                if (lane < nb) {
                    my_b = P.A.rowptr[P.A.rows[ri0 + lane]];
                    my_w = P.rlen[ri0 + lane];
                }
                auto row_start = [&](uint32_t q) -> const uint32_t * {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_b, (int)q);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_b >> 32), (int)q);
                    return P.A.col + (((unsigned long long)hi << 32) | lo);
                };
                {   // the batch's short rows (all of stream BN): lane r fetches row r's columns, all loads in flight at once, into LDS
                    uint32_t v[S3_BCOLS];
                    const uint32_t *mine = P.A.col + my_b;
#pragma unroll
                    for (uint32_t tt = 0; tt < S3_BCOLS; ++tt) v[tt] = (my_w <= S3_BCOLS && tt < my_w) ? mine[tt] : 0u;
#pragma unroll
                    for (uint32_t tt = 0; tt < S3_BCOLS; ++tt) scol[lane][tt] = v[tt];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                // (two bodies: a short row's columns come from LDS and its body holds no global LOAD -- with one in it the compiler
                // waits with vmcnt(0) at the top of every row, i.e. for the previous row's dictionary stores, ~1 us)
                auto do_row = [&](auto short_tag, uint32_t q, uint32_t w) {
                    constexpr bool SHORT = decltype(short_tag)::value;
                    const uint32_t ri = ri0 + q;
                    const uint32_t *set = SHORT ? nullptr : row_start(q);
                    const uint32_t c0 = SHORT ? scol[q][lane < S3_BCOLS ? lane : 0u] : 0u;
                    bool fr0 = false;  // (short rows: the first pass's answer serves the second -- a tile closed in between empties the set)
                    for (;;) {
                        uint32_t fresh = 0;
                        for (uint32_t base = 0; base < (SHORT ? 1u : w); base += 64) {
                            const uint32_t i = base + lane;
                            const uint32_t c = SHORT ? c0 : (i < w ? set[i] : 0u);
                            fr0 = i < w && !tab_has(c);
                            fresh += (uint32_t)__popcll(__ballot(fr0));
                        }
                        if (tile_cols + fresh <= (uint32_t)PSELL_TILE_COLS_TARGET || (tile_cols == 0 && in_slice == 0 && pending == 0)) break;
                        close_slice();
                        close_tile();
                    }
                    for (uint32_t base = 0; base < (SHORT ? 1u : w); base += 64) {
                        const uint32_t i = base + lane;
                        const uint32_t c = SHORT ? c0 : (i < w ? set[i] : 0u);
                        const bool fr = SHORT ? fr0 : (i < w && !tab_has(c));
                        const uint64_t bal = __ballot(fr);
                        if (fr) {
                            tab_put(c);
                            dict_out[dict_n + (uint32_t)__popcll(bal & lanes_below())] = c;
                        }
                        const uint32_t cnt = (uint32_t)__popcll(bal);
                        dict_n += cnt;
                        tile_cols += cnt;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (LDS only: a wave's LDS operations complete in order)
                    }
                    if (in_slice == 0) slice_start = ri;
                    ++in_slice;
                    if (in_slice == (uint32_t)PSELL_LANES) {
                        close_slice();
                        if (pending >= cap) close_tile();
                    }
                };
                for (uint32_t q = 0; q < nb; ++q) {
                    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)my_w, (int)q);
                    if (w <= S3_BCOLS) do_row(std::true_type{}, q, w);
                    else do_row(std::false_type{}, q, w);
                }
            }
            close_slice();
            close_tile();
        }
        if (lane == 0) P.outs[k] = SegOut{ntile, nsl, dict_n, (uint32_t)(wall_clock64() - t_seg0)};
    }
}

struct SegFinal {
    uint32_t tile_base, slice_base, dict_base, pad;
};

// the tiles of every segment into the final numbering; the dictionaries copied (padding entries stay 0)
__global__ void s3_tiles_final_kernel(const SegDesc *segs, const SegAux *aux, const SegOut *outs, const SegFinal *fin, uint32_t nseg,
                                      const uint32_t *t_s0, const uint32_t *t_cols, const uint32_t *t_dstart, const uint32_t *dict_s,
                                      uint32_t *tile_slice, uint32_t *tile_dict, uint32_t *tile_cols, uint32_t *tile_seg, uint32_t *dict)
{
    const uint32_t k = blockIdx.x;
    if (k >= nseg) return;
    const SegAux x = aux[k];
    const SegFinal f = fin[k];
    const uint32_t nt = outs[k].ntiles;
    for (uint32_t t = threadIdx.x >> 6; t < nt; t += blockDim.x >> 6) {  // one wave per tile
        const uint32_t d0 = t_dstart[x.tbase + t], L = t_cols[x.tbase + t];
        if (lane_id() == 0) {
            tile_slice[f.tile_base + t] = f.slice_base + t_s0[x.tbase + t];
            tile_dict[f.tile_base + t] = f.dict_base + d0;
            tile_cols[f.tile_base + t] = L;
            tile_seg[f.tile_base + t] = k;
        }
        for (uint32_t l = lane_id(); l < L; l += 64) dict[f.dict_base + d0 + l] = dict_s[x.dbase + d0 + l];
    }
}

struct S3Size {
    S3In A;
    const SegDesc *segs;
    const SegAux *aux;
    const SegFinal *fin;
    const uint32_t *rlen, *endpos, *srec_ri, *srec_n;
    const uint32_t *tile_slice, *tile_seg;
    uint32_t num_tiles;
    int has_ks;
    uint32_t *sl_ri, *sl_n, *sl_units, *sl_tile, *sl_long;
    uint8_t *slice_flags, *slice_w;
    unsigned long long *stats;  // [3][8] rows, nnz, bytes per stream; [24] padded_nnz
};

struct SliceSet {
    const uint32_t *set;
    uint32_t w;
};
__device__ inline void slice_rows_of(const S3Size &P, const SegDesc &seg, const SegAux &x, const SegFinal &f, uint32_t s, uint32_t &rs, uint32_t &nr)
{
    const uint32_t ls = s - f.slice_base;
    if (seg.stream <= PSELL_A2M) {
        const uint32_t j = x.sl0 + ls;
        rs = j == 0 ? 0u : P.endpos[j - 1] + 1u;
        nr = P.endpos[j] + 1u - rs;
    } else {
        const uint32_t sb = seg.ra - P.A.bounds[PSELL_BN];
        rs = P.srec_ri[sb + ls];
        nr = P.srec_n[sb + ls];
    }
}
__device__ inline SliceSet uniform_set(const S3In &A, const uint32_t *rlen, uint32_t rs)
{
    SliceSet q;
    if (A.form[rs]) {
        const uint32_t g = A.gid[rs];
        q.set = A.pat_col + A.pat_ptr[g];
        q.w = A.pat_ptr[g + 1] - A.pat_ptr[g];
    } else {
        q.set = A.col + A.rowptr[A.rows[rs]];
        q.w = rlen[rs];
    }
    return q;
}

// per slice (one wave per tile, a lane per slice): its rows, width, bytes and flags
__global__ __launch_bounds__(64) void s3_size_kernel(S3Size P)
{
    // (the totals: summed per tile in LDS, one global atomic per tile and counter -- every slice adding to the same four global words was
    // 2.4 M same-address atomics at C2, most of this kernel's 8.7 ms)
    __shared__ unsigned long long acc[25];
    const uint32_t t = blockIdx.x;
    if (t >= P.num_tiles) return;
    if (lane_id() < 25) acc[lane_id()] = 0ull;
    __syncthreads();
    const uint32_t k = P.tile_seg[t];
    const SegDesc seg = P.segs[k];
    const SegAux x = P.aux[k];
    const SegFinal f = P.fin[k];
    const int st = (int)seg.stream;
    const bool uniform = st <= PSELL_A2M;
    const uint32_t s0 = P.tile_slice[t], s1 = P.tile_slice[t + 1];
    for (uint32_t s = s0 + lane_id(); s < s1; s += 64) {
        uint32_t rs, nr;
        slice_rows_of(P, seg, x, f, s, rs, nr);
        uint32_t longest = 0;
        unsigned long long nnz = 0;
        for (uint32_t q = 0; q < nr; ++q) {
            const uint32_t l = P.rlen[rs + q];
            longest = max(longest, l);
            nnz += l;
        }
        uint32_t w = longest, form = 0;
        uint8_t flags = 0;
        if (uniform) {
            form = P.A.form[rs];
            const SliceSet q = uniform_set(P.A, P.rlen, rs);
            w = q.w;
            flags |= 1;
            if (s > s0) {  // the same set as the previous slice of this tile?
                uint32_t prs, pnr;
                slice_rows_of(P, seg, x, f, s - 1, prs, pnr);
                const SliceSet p = uniform_set(P.A, P.rlen, prs);
                bool same = p.w == q.w;
                for (uint32_t i = 0; same && i < q.w; ++i) same = p.set[i] == q.set[i];
                if (same) flags |= 2;
            }
        } else {
            const uint32_t r0 = P.A.rows[rs];
            const uint64_t b0 = P.A.rowptr[r0];
            const uint32_t len0 = P.rlen[rs];
            bool uni = true;
            for (uint32_t q = 1; uni && q < nr; ++q) {
                const uint64_t b = P.A.rowptr[P.A.rows[rs + q]];
                uni = P.rlen[rs + q] == len0;
                for (uint32_t i = 0; uni && i < len0; ++i) uni = P.A.col[b + i] == P.A.col[b0 + i];
            }
            if (uni) flags |= 1;
        }
        const bool masked = st == PSELL_A1M || st == PSELL_A2M || (st == PSELL_A1 && form == 2);
        if (masked) flags |= 4;
        uint32_t bytes, stored = w;
        if (masked) {
            stored = longest;
            bytes = 256u * (st == PSELL_A2M ? 2u : 1u) + longest * 256u + (P.has_ks ? 256u : 0u);
        } else if (uniform) {
            bytes = 256u + w * 256u + (P.has_ks ? 256u : 0u);
        } else {
            bytes = ((w * 384u + 255u) & ~255u) + (P.has_ks && st == PSELL_BN ? 256u : 0u);
        }
        P.sl_ri[s] = rs;
        P.sl_n[s] = nr;
        P.sl_units[s] = bytes / 128u;
        P.sl_tile[s] = t;
        P.sl_long[s] = longest;
        P.slice_flags[s] = flags;
        P.slice_w[s] = (uint8_t)min(w, 255u);
        const int ss = masked && st == PSELL_A1 ? PSELL_A1M : st;
        atomicAdd(&acc[ss], (unsigned long long)nr);
        atomicAdd(&acc[8 + ss], nnz);
        atomicAdd(&acc[16 + ss], (unsigned long long)bytes);
        atomicAdd(&acc[24], (unsigned long long)stored * 64ull);
    }
    __syncthreads();
    if (lane_id() < 25 && acc[lane_id()] != 0ull) atomicAdd(&P.stats[lane_id()], acc[lane_id()]);
}

struct S3Emit {
    S3In A;
    const SegDesc *segs;
    const uint32_t *tile_seg, *tile_dict, *tile_cols, *dict;
    const uint32_t *sl_ri, *sl_n, *sl_tile, *sl_long, *slice_off;
    uint32_t num_slices;
    uint8_t *data;
    uint32_t *row_order;  // or null
    float *slice_ks;      // or null
};

__device__ inline uint32_t local_id(const uint32_t *sdict, uint32_t L, uint32_t c)
{
    uint32_t i = 0;
    while (i < L && sdict[i] != c) ++i;
    return i;
}

// THE BYTES: one wave per slice, a lane per fragment (psell_build.cpp, emit_slice; `data` arrives zeroed)
__global__ __launch_bounds__(64) void s3_emit_kernel(S3Emit E)
{
    __shared__ uint32_t sdict[PSELL_MAX_TILE_COLS];
    __shared__ uint32_t spat[PSELL_WIDE_MAX];
    __shared__ uint32_t spat_local[PSELL_WIDE_MAX];
    const uint32_t s = blockIdx.x;
    if (s >= E.num_slices) return;
    const uint32_t lane = lane_id();
    const uint32_t t = E.sl_tile[s];
    const int st = (int)E.segs[E.tile_seg[t]].stream;
    const uint32_t d0 = E.tile_dict[t], L = E.tile_cols[t];
    for (uint32_t i = lane; i < L && i < (uint32_t)PSELL_MAX_TILE_COLS; i += 64) sdict[i] = E.dict[d0 + i];
    const uint32_t rs = E.sl_ri[s], nr = E.sl_n[s], longest = E.sl_long[s];
    const bool uniform = st <= PSELL_A2M;
    const uint32_t form = uniform ? E.A.form[rs] : 0u;
    const bool masked = st == PSELL_A1M || st == PSELL_A2M || (st == PSELL_A1 && form == 2);
    uint8_t *base = E.data + (size_t)E.slice_off[s] * 128u;
    const bool valid = lane < nr;
    const uint32_t r = valid ? E.A.rows[rs + lane] : 0u;
    const uint64_t b = valid ? E.A.rowptr[r] : 0ull;
    const uint32_t len = valid ? (uint32_t)(E.A.rowptr[r + 1] - b) : 0u;
    uint32_t w = longest;
    if (uniform) {
        const uint32_t *set;
        if (form) {
            const uint32_t g = E.A.gid[rs];
            set = E.A.pat_col + E.A.pat_ptr[g];
            w = E.A.pat_ptr[g + 1] - E.A.pat_ptr[g];
        } else {
            set = E.A.col + E.A.rowptr[E.A.rows[rs]];
            w = (uint32_t)(E.A.rowptr[E.A.rows[rs] + 1] - E.A.rowptr[E.A.rows[rs]]);
        }
        __syncthreads();
        if (lane < w) {
            spat[lane] = set[lane];
            spat_local[lane] = local_id(sdict, L, set[lane]);
        }
    }
    __syncthreads();
    const float ksv = E.A.ks && valid ? (float)E.A.ks[r] : 0.0f;
    if (masked) {
        const uint32_t hrows = st == PSELL_A2M ? 2u : 1u;
        uint32_t *hw = reinterpret_cast<uint32_t *>(base);
        float *vals = reinterpret_cast<float *>(base + 256u * hrows);
        uint32_t mk = 0, tpos = 0;
        for (uint32_t k = 0; k < len; ++k) {
            const uint32_t c = E.A.col[b + k];
            while (spat[tpos] != c) ++tpos;
            mk |= 1u << tpos;
            vals[(size_t)k * 64 + lane] = E.A.val[b + k];
        }
        for (uint32_t h = 0; h < hrows; ++h) {
            uint32_t word = 0;
            if (lane < 16) {
                const uint32_t tt = 16u * h + lane;
                word = (uint32_t)(tt < w ? spat_local[tt] : (uint32_t)PSELL_NO_COL) << 16;
            }
            if (valid) word |= h == 0 ? (mk & 0xffffu) : (mk >> 16);
            hw[h * 64 + lane] = word;
        }
        if (E.A.ks && valid) reinterpret_cast<float *>(base + 256u * hrows + (size_t)longest * 256u)[lane] = ksv;
    } else if (uniform) {
        uint16_t *hdr = reinterpret_cast<uint16_t *>(base);
        float *vals = reinterpret_cast<float *>(base + 256);
        if (lane < w) hdr[lane] = (uint16_t)spat_local[lane];
        if (valid) {
            if (form == 0) {
                for (uint32_t tt = 0; tt < w; ++tt) vals[(size_t)tt * 64 + psell_row_pos(st, tt, lane)] = E.A.val[b + tt];
            } else {
                uint32_t tpos = 0;
                for (uint32_t k = 0; k < len; ++k) {
                    const uint32_t c = E.A.col[b + k];
                    while (spat[tpos] != c) ++tpos;
                    vals[(size_t)tpos * 64 + psell_row_pos(st, tpos, lane)] = E.A.val[b + k];
                }
            }
            if (E.A.ks) reinterpret_cast<float *>(base + 256 + (size_t)w * 256u)[lane] = ksv;
        }
    } else {
        const size_t body = ((size_t)w * 384 + 255) & ~(size_t)255;
        float *vals = reinterpret_cast<float *>(base);
        uint16_t *lcols = reinterpret_cast<uint16_t *>(base + (size_t)w * 256);
        if (valid) {
            uint16_t last = 0;
            for (uint32_t tt = 0; tt < w; ++tt) {
                if (tt < len) {
                    vals[(size_t)tt * 64 + lane] = E.A.val[b + tt];
                    last = (uint16_t)local_id(sdict, L, E.A.col[b + tt]);
                }
                lcols[(size_t)tt * 64 + lane] = last;
            }
            if (E.A.ks && st == PSELL_BN) reinterpret_cast<float *>(base + body)[lane] = ksv;
        }
    }
    if (E.row_order) E.row_order[(size_t)s * 64 + lane] = valid ? r : 0xffffffffu;
    if (E.slice_ks) E.slice_ks[(size_t)s * 64 + lane] = ksv;
}

__global__ void s3_flags_kernel(uint32_t S, const uint8_t *slice_flags, uint32_t *slice_off)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    slice_off[s] |= ((uint32_t)(slice_flags[s] & 3u) << 30) | ((uint32_t)((slice_flags[s] >> 2) & 1u) << PSELL_FLAG_MASKED_BIT);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
typedef DevScratch Scratch;  // (common.hpp: a kept block; trims and retries when memory is short)

template <typename In, typename Out, typename T>
hipError_t exclusive_sum(Scratch &tmp, In in, Out out, T init, size_t count, hipStream_t stream)
{
    size_t bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, init, count, rocprim::plus<T>(), stream);
    if (e != hipSuccess) return e;
    if ((e = tmp.need(bytes)) != hipSuccess) return e;
    return rocprim::exclusive_scan(tmp.p, bytes, in, out, init, count, rocprim::plus<T>(), stream);
}

struct ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

}  // namespace


// ================================================================= STAGE 1 ==================================================
namespace {

__device__ inline uint32_t mix32_dev(uint32_t h, uint32_t v)  // (psell_build.cpp, mix32)
{
    h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2);
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}

struct S1Counters {
    unsigned long long empties, singles;
    uint32_t err, max_row;
};

// sort key of every row (first transcript's bin, length, hash of the set); empty rows and rows with ONE transcript drop out --
// the latter counted per transcript (stream S), their k log X_ij left in `term`
__global__ __launch_bounds__(256) void s1_keys_kernel(PsellDevIn X, int binsh, uint64_t *keys, uint32_t *keep, uint32_t *is_single, double *term,
                                                     unsigned long long *scnt, S1Counters *ctr)
{
    // (the counters: per thread, then per block, then ONE global atomic per block and counter -- a fixed grid walks the rows.  Every
    // row's atomicMax / atomicAdd on the same four words, even wave-aggregated, was most of this kernel's 7.3 ms at C2.)
    __shared__ unsigned long long sh_cnt[2];
    __shared__ uint32_t sh_u[2];
    if (threadIdx.x < 2) {
        sh_cnt[threadIdx.x] = 0ull;
        sh_u[threadIdx.x] = 0u;
    }
    __syncthreads();
    uint32_t my_err = 0, my_max = 0, n_empty = 0, n_single = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)X.m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = X.rowptr[i], e = X.rowptr[i + 1];
        uint64_t key = ~0ull;
        uint32_t single = 0;
        double tm = 0.0;
        if (e < b) {
            my_err = max(my_err, 1u);
        } else if (e == b) {
            ++n_empty;
        } else if (e - b > (uint64_t)PSELL_MAX_TILE_COLS) {
            my_err = max(my_err, 2u);
        } else if (e - b == 1 && X.col[b] < (uint64_t)X.n && X.val[b] > 0.0f && (!X.ks || X.ks[i] >= 0)) {
            const long long k = X.ks ? X.ks[i] : 1;
            atomicAdd(&scnt[X.col[b]], (unsigned long long)k);
            tm = (double)k * log((double)X.val[b]);
            single = 1;
            ++n_single;
            my_max = max(my_max, 1u);
        } else {
            const uint64_t len = e - b;
            uint32_t h = 0x12345u;
            const uint32_t first = X.col[b];
            for (uint64_t k = b; k < e; ++k) {
                const uint32_t c = X.col[k];
                if (c >= (uint64_t)X.n) my_err = max(my_err, 3u);
                if (k > b && c <= X.col[k - 1]) my_err = max(my_err, 4u);
                h = mix32_dev(h, c);
            }
            my_max = max(my_max, (uint32_t)len);
            key = ((uint64_t)(first >> binsh) << 40) | ((uint64_t)(len < 255 ? len : 255) << 32) | h;
        }
        keys[i] = key;
        keep[i] = key != ~0ull;
        is_single[i] = single;
        term[i] = tm;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        my_err = max(my_err, (uint32_t)__shfl_xor((int)my_err, d, 64));
        my_max = max(my_max, (uint32_t)__shfl_xor((int)my_max, d, 64));
        n_empty += (uint32_t)__shfl_xor((int)n_empty, d, 64);
        n_single += (uint32_t)__shfl_xor((int)n_single, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (my_err) atomicMax(&sh_u[0], my_err);
        if (my_max) atomicMax(&sh_u[1], my_max);
        if (n_empty) atomicAdd(&sh_cnt[0], (unsigned long long)n_empty);
        if (n_single) atomicAdd(&sh_cnt[1], (unsigned long long)n_single);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sh_u[0]) atomicMax(&ctr->err, sh_u[0]);
        if (sh_u[1]) atomicMax(&ctr->max_row, sh_u[1]);
        if (sh_cnt[0]) atomicAdd(&ctr->empties, sh_cnt[0]);
        if (sh_cnt[1]) atomicAdd(&ctr->singles, sh_cnt[1]);
    }
}

__global__ void s1_compact_kernel(uint64_t m, const uint64_t *keys, const uint32_t *keep, const uint32_t *kscan, const uint32_t *is_single,
                                  const uint32_t *sscan, uint64_t *keys_c, uint32_t *rows_c, uint32_t *single_rows)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (keep[i]) {
        keys_c[kscan[i]] = keys[i];
        rows_c[kscan[i]] = (uint32_t)i;
    }
    if (is_single[i]) single_rows[sscan[i]] = (uint32_t)i;
}

__global__ void s1_heads_kernel(PsellDevIn X, uint32_t N, const uint32_t *rows, uint32_t *head)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t h = 1;
    if (i > 0) {
        const uint32_t r1 = rows[i - 1], r2 = rows[i];
        const uint64_t b1 = X.rowptr[r1], b2 = X.rowptr[r2];
        const uint64_t l1 = X.rowptr[r1 + 1] - b1, l2 = X.rowptr[r2 + 1] - b2;
        bool same = l1 == l2;
        for (uint64_t k = 0; same && k < l1; ++k) same = X.col[b1 + k] == X.col[b2 + k];
        h = same ? 0u : 1u;
    }
    head[i] = h;
}

__global__ void s1_runstart_kernel(uint32_t N, const uint32_t *head, const uint32_t *hscan, uint32_t *run_start)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    if (i == N) {
        run_start[hscan[N]] = N;
        return;
    }
    if (head[i]) run_start[hscan[i]] = i;
}

// a run of r identical rows gives floor(r / 64) whole slices and its remainder as one more when that is >= 32 rows
// (psell_build.cpp, "exact runs -> streams"); the other rows are leftover
__global__ void s1_classify_kernel(PsellDevIn X, uint32_t N, const uint32_t *rows, const uint32_t *head, const uint32_t *hscan,
                                   const uint32_t *run_start, uint32_t *f1, uint32_t *f2, uint32_t *fb, uint8_t *endf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t rid = hscan[i] + head[i] - 1u;
    const uint32_t s = run_start[rid], e = run_start[rid + 1];
    const uint32_t r = e - s, q = i - s;
    const uint32_t row = rows[i];
    const uint64_t len = X.rowptr[row + 1] - X.rowptr[row];
    uint32_t take = (r / PSELL_LANES) * PSELL_LANES;
    if (r - take >= (uint32_t)PSELL_MIN_UNIFORM_ROWS) take = r;
    if (len > (uint64_t)PSELL_WIDE_MAX) take = 0;
    const bool taken = q < take;
    const bool narrow = len <= (uint64_t)PSELL_NARROW_MAX;
    f1[i] = taken && narrow;
    f2[i] = taken && !narrow;
    fb[i] = !taken;
    endf[i] = taken && ((q + 1) % PSELL_LANES == 0 || q + 1 == take) ? 1 : 0;
}

__global__ void s1_split_kernel(uint32_t N, const uint32_t *rows, const uint32_t *f1, const uint32_t *f2, const uint32_t *fb,
                                const uint32_t *p1, const uint32_t *p2, const uint32_t *pb, const uint8_t *endf, uint32_t *a1_rows,
                                uint32_t *a1_ends, uint32_t *a2_rows, uint32_t *a2_ends, uint32_t *rb)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t row = rows[i];
    if (f1[i]) {
        a1_rows[p1[i]] = row;
        a1_ends[p1[i]] = endf[i];
    } else if (f2[i]) {
        a2_rows[p2[i]] = row;
        a2_ends[p2[i]] = endf[i];
    } else if (fb[i]) {
        rb[pb[i]] = row;
    }
}

__global__ void s1_cnt_to_float_kernel(uint32_t n, const unsigned long long *scnt, float *out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = (float)(long long)scnt[j];
}

}  // namespace

#define PD_HIP(expr) POLEE_HIP_TRY(ctx, expr)

polee_status psell_device_stage1(polee_ctx *ctx, const PsellDevIn &X, PsellHost &out, PsellDevRuns &R, bool want_debug)
{
    hipStream_t stream = ctx->stream;
    const int64_t m = X.m, n = X.n;
    if (m < 0 || n < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: bad matrix dimensions");
    if (n > (int64_t)1 << 31) return fail(ctx, POLEE_ERR_UNSUPPORTED, "likelihood matrix: more than 2^31 transcripts is not supported");
    if (m >= ((int64_t)1 << 32) - 1) return fail(ctx, POLEE_ERR_UNSUPPORTED, "likelihood matrix: more than 2^32 fragments is not supported");
    out = PsellHost();
    out.m = m;
    out.n = n;
    R.n_a1 = R.n_a2 = R.n_rb = 0;
    uint64_t nnz = 0;
    PD_HIP(hipMemcpyAsync(&nnz, X.rowptr + m, 8, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    out.nnz = (int64_t)nnz;
    if (m == 0) return POLEE_OK;
    const int binsh = psell_bin_shift();
    const unsigned TB = 256;
    Scratch tmp(ctx);
    DevBuf<uint64_t> keys, keys_c, keys_s;
    DevBuf<uint32_t> keep, is_single, kscan, sscan, rows_c, rows_s, single_rows;
    DevBuf<double> term, lsum;
    DevBuf<unsigned long long> scnt;
    DevBuf<S1Counters> ctr;
    POLEE_TRY(keys.alloc(ctx, (size_t)m));
    POLEE_TRY(keep.alloc(ctx, (size_t)m + 1));
    POLEE_TRY(is_single.alloc(ctx, (size_t)m + 1));
    POLEE_TRY(kscan.alloc(ctx, (size_t)m + 1));
    POLEE_TRY(sscan.alloc(ctx, (size_t)m + 1));
    POLEE_TRY(term.alloc(ctx, (size_t)m));
    POLEE_TRY(lsum.alloc(ctx, 1));
    POLEE_TRY(scnt.alloc(ctx, (size_t)n));
    POLEE_TRY(ctr.alloc(ctx, 1));
    PD_HIP(hipMemsetAsync(scnt.p, 0, (size_t)n * 8, stream));
    PD_HIP(hipMemsetAsync(ctr.p, 0, sizeof(S1Counters), stream));
    PD_HIP(hipMemsetAsync(keep.p + m, 0, 4, stream));
    PD_HIP(hipMemsetAsync(is_single.p + m, 0, 4, stream));
    hipLaunchKernelGGL(s1_keys_kernel, dim3((unsigned)std::min<uint64_t>(((uint64_t)m + TB - 1) / TB, 8192)), dim3(TB), 0, stream, X, binsh, keys.p, keep.p, is_single.p, term.p,
                       scnt.p, ctr.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, keep.p, kscan.p, 0u, (size_t)m + 1, stream));
    PD_HIP(exclusive_sum(tmp, is_single.p, sscan.p, 0u, (size_t)m + 1, stream));
    {
        size_t bytes = 0;
        PD_HIP(rocprim::reduce(nullptr, bytes, term.p, lsum.p, 0.0, (size_t)m, rocprim::plus<double>(), stream));
        PD_HIP(tmp.need(bytes));
        PD_HIP(rocprim::reduce(tmp.p, bytes, term.p, lsum.p, 0.0, (size_t)m, rocprim::plus<double>(), stream));
    }
    S1Counters h_ctr;
    uint32_t N = 0, nsingle = 0;
    double h_lsum = 0.0;
    PD_HIP(hipMemcpyAsync(&h_ctr, ctr.p, sizeof h_ctr, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(&N, kscan.p + m, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(&nsingle, sscan.p + m, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(&h_lsum, lsum.p, 8, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (h_ctr.err == 1) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: row offsets are not monotone");
    if (h_ctr.err == 2) return fail(ctx, POLEE_ERR_UNSUPPORTED, "likelihood matrix: a fragment is compatible with more than 1024 transcripts");
    if (h_ctr.err == 3) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: transcript index out of range");
    if (h_ctr.err == 4) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: the transcript ids of a fragment must be strictly ascending (sorted, no duplicates)");
    out.empty_rows = (int64_t)h_ctr.empties;
    out.max_row = (int32_t)h_ctr.max_row;
    POLEE_TRY(keys_c.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(rows_c.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(single_rows.alloc(ctx, (size_t)nsingle + 1));
    hipLaunchKernelGGL(s1_compact_kernel, dim3((unsigned)((m + TB - 1) / TB)), dim3(TB), 0, stream, (uint64_t)m, keys.p, keep.p, kscan.p, is_single.p,
                       sscan.p, keys_c.p, rows_c.p, single_rows.p);
    POLEE_KERNEL_CHECK(ctx);
    if (nsingle > 0) {
        POLEE_TRY(R.single_cnt.alloc(ctx, (size_t)n));
        hipLaunchKernelGGL(s1_cnt_to_float_kernel, dim3((unsigned)((n + TB - 1) / TB)), dim3(TB), 0, stream, (uint32_t)n, scnt.p, R.single_cnt.p);
        POLEE_KERNEL_CHECK(ctx);
        out.single_cnt.resize((size_t)n);
        PD_HIP(hipMemcpyAsync(out.single_cnt.data(), R.single_cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        out.single_logsum = h_lsum;
        out.stream_rows[PSELL_S] = nsingle;
        out.stream_nnz[PSELL_S] = nsingle;
        out.stream_bytes[PSELL_S] = 4 * n;
        if (want_debug) {
            out.single_rows.resize(nsingle);
            PD_HIP(hipMemcpyAsync(out.single_rows.data(), single_rows.p, (size_t)nsingle * 4, hipMemcpyDeviceToHost, stream));
        }
        PD_HIP(hipStreamSynchronize(stream));
    }
    keys.release();
    term.release();
    if (N == 0) return POLEE_OK;
    // stable sort by key (the host's LSD radix sort is stable too: the same order)
    POLEE_TRY(keys_s.alloc(ctx, N));
    POLEE_TRY(rows_s.alloc(ctx, (size_t)N + 1));
    {
        size_t bytes = 0;
        PD_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_c.p, keys_s.p, rows_c.p, rows_s.p, (size_t)N, 0, 64, stream));
        PD_HIP(tmp.need(bytes));
        PD_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, keys_c.p, keys_s.p, rows_c.p, rows_s.p, (size_t)N, 0, 64, stream));
    }
    keys_c.release();
    keys_s.release();
    DevBuf<uint32_t> head, hscan, run_start, f1, f2, fb, p1, p2, pb;
    DevBuf<uint8_t> endf;
    POLEE_TRY(head.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(hscan.alloc(ctx, (size_t)N + 1));
    PD_HIP(hipMemsetAsync(head.p + N, 0, 4, stream));
    hipLaunchKernelGGL(s1_heads_kernel, dim3((N + TB - 1) / TB), dim3(TB), 0, stream, X, N, rows_s.p, head.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, head.p, hscan.p, 0u, (size_t)N + 1, stream));
    uint32_t nruns = 0;
    PD_HIP(hipMemcpyAsync(&nruns, hscan.p + N, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    POLEE_TRY(run_start.alloc(ctx, (size_t)nruns + 1));
    hipLaunchKernelGGL(s1_runstart_kernel, dim3((N + 1 + TB - 1) / TB), dim3(TB), 0, stream, N, head.p, hscan.p, run_start.p);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(f1.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(f2.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(fb.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(p1.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(p2.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(pb.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(endf.alloc(ctx, (size_t)N + 1));
    PD_HIP(hipMemsetAsync(f1.p + N, 0, 4, stream));
    PD_HIP(hipMemsetAsync(f2.p + N, 0, 4, stream));
    PD_HIP(hipMemsetAsync(fb.p + N, 0, 4, stream));
    hipLaunchKernelGGL(s1_classify_kernel, dim3((N + TB - 1) / TB), dim3(TB), 0, stream, X, N, rows_s.p, head.p, hscan.p, run_start.p, f1.p, f2.p,
                       fb.p, endf.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, f1.p, p1.p, 0u, (size_t)N + 1, stream));
    PD_HIP(exclusive_sum(tmp, f2.p, p2.p, 0u, (size_t)N + 1, stream));
    PD_HIP(exclusive_sum(tmp, fb.p, pb.p, 0u, (size_t)N + 1, stream));
    uint32_t c1 = 0, c2 = 0, cb = 0;
    PD_HIP(hipMemcpyAsync(&c1, p1.p + N, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(&c2, p2.p + N, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(&cb, pb.p + N, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    R.n_a1 = c1;
    R.n_a2 = c2;
    R.n_rb = cb;
    POLEE_TRY(R.a1_rows.alloc(ctx, (size_t)c1 + 1));
    POLEE_TRY(R.a1_ends.alloc(ctx, (size_t)c1 + 1));
    POLEE_TRY(R.a2_rows.alloc(ctx, (size_t)c2 + 1));
    POLEE_TRY(R.a2_ends.alloc(ctx, (size_t)c2 + 1));
    POLEE_TRY(R.rb.alloc(ctx, (size_t)cb + 1));
    hipLaunchKernelGGL(s1_split_kernel, dim3((N + TB - 1) / TB), dim3(TB), 0, stream, N, rows_s.p, f1.p, f2.p, fb.p, p1.p, p2.p, pb.p, endf.p,
                       R.a1_rows.p, R.a1_ends.p, R.a2_rows.p, R.a2_ends.p, R.rb.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(hipStreamSynchronize(stream));
    return POLEE_OK;
}

polee_status psell_device_runs_to_host(polee_ctx *ctx, const PsellDevRuns &R, PsellRuns &H)
{
    H = PsellRuns();
    H.a1_rows.resize(R.n_a1); H.a1_ends.resize(R.n_a1); H.a2_rows.resize(R.n_a2); H.a2_ends.resize(R.n_a2); H.rb.resize(R.n_rb);
    POLEE_TRY(R.a1_rows.download(ctx, H.a1_rows.data(), R.n_a1));
    POLEE_TRY(R.a1_ends.download(ctx, H.a1_ends.data(), R.n_a1));
    POLEE_TRY(R.a2_rows.download(ctx, H.a2_rows.data(), R.n_a2));
    POLEE_TRY(R.a2_ends.download(ctx, H.a2_ends.data(), R.n_a2));
    POLEE_TRY(R.rb.download(ctx, H.rb.data(), R.n_rb));
    return POLEE_OK;
}

// ================================================================= STAGE 2 ==================================================
namespace {

constexpr uint32_t S2_CHUNK = PSELL_PACK_CHUNK;  // candidate rows packed independently (psell_build.cpp: UCH)
constexpr uint32_t S2_MAX_GROUP = 1u << 14;

__device__ inline uint32_t wave_sum_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline uint32_t wave_max_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

__global__ void s2_firstcol_kernel(PsellDevIn X, uint32_t N, const uint32_t *rows, uint32_t *key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) key[i] = X.col[X.rowptr[rows[i]]];
}

// class of a row by its length: 0 (<= t0), 1 (<= t1), 2
__global__ void s2_lenclass_kernel(PsellDevIn X, uint32_t N, const uint32_t *rows, uint32_t t0, uint32_t t1, uint8_t *cls)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint64_t len = X.rowptr[rows[i] + 1] - X.rowptr[rows[i]];
    cls[i] = len <= t0 ? 0 : (len <= t1 ? 1 : 2);
}
__global__ void s2_flag_kernel(uint32_t N, const uint8_t *cls, uint8_t k, uint32_t mask, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flag[i] = (cls[i] & mask) == k;
}
__global__ void s2_scatter_rows_kernel(uint32_t N, const uint32_t *flag, const uint32_t *pos, const uint32_t *rows, uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && flag[i]) out[pos[i]] = rows[i];
}

__global__ void s2_poolinfo_kernel(PsellDevIn X, uint32_t N, const uint32_t *prow, uint32_t *plen, uint64_t *pbeg)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint64_t b = X.rowptr[prow[i]];
    pbeg[i] = b;
    plen[i] = (uint32_t)(X.rowptr[prow[i] + 1] - b);
}

// LDS hand-over between the lanes of ONE wave (the block is one wave): DS operations of a wave execute in order, so only the
// compiler has to be kept from moving them -- no wait on the vector-memory counter, which would also wait for the row that is
// being prefetched
#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

__device__ inline uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

struct S2Pack {
    const uint32_t *col;
    const uint32_t *prow, *plen;
    const uint64_t *pbeg, *plps;
    uint32_t Np, cap;
    int pass_w, ks_rows;
    double factor;
    uint32_t *rec_row, *rec_gid, *chunk_npats, *psize, *poff, *pcols, *gbuf, *dbuf;
    uint8_t *rec_meta;  // dest (bits 0..2: 0 dense narrow, 1 masked narrow, 2 dense wide, 3 masked wide, 4 left) | end << 3 | form << 4
};

// PACKING OF THE LEFTOVER ROWS (psell_build.cpp, stage 2: "Rows are visited in the order of their first transcript ... a row
// joins the open group while the union stays within the pass's width ... a misfit is deferred once"): the same greedy walk,
// one wave per chunk of 32 768 candidates, the union held sorted in LDS, a lane per transcript of the row.
__global__ __launch_bounds__(64) void s2_pack_kernel(S2Pack P)
{
#pragma clang fp contract(off)
    __shared__ uint32_t smerge[PSELL_LANES];
    constexpr uint32_t GL = 256;  // rows of the open group held in LDS (id and length); longer groups continue in global memory
    __shared__ uint32_t g_row[GL], g_len[GL];
    const uint32_t lane = lane_id();
    const uint32_t p0 = blockIdx.x * S2_CHUNK, p1 = min(P.Np, p0 + S2_CHUNK);
    uint32_t u_reg = 0xffffffffu;  // the open group's union, ascending: lane j holds its j-th transcript (j < nu)
    uint32_t nu = 0, gs = 0, emitted = 0, npat = 0, pat_off = 0, nd = 0;
    const double relax = 2.0, relax0 = 2.0, mask_gain = 0.5;
    double allowance = (8.0 * (double)(P.plps[p1] - P.plps[p0]) + 4.0 * (double)(p1 - p0)) * P.factor;
    double wide_allowance = P.pass_w == 1 ? PSELL_PACK_WIDE_RESERVE : 0.0;
    const uint64_t pat_base = P.plps[p0];
    auto close_group = [&]() {
        if (gs == 0) return;
        if (gs > GL) __syncthreads();  // (the group's rows, written by lane 0, to every lane)
        else WAVE_LDS_SYNC();
        const bool narrow = nu <= (uint32_t)PSELL_NARROW_MAX;
        const uint32_t pid = npat;
        bool any = false;
        for (uint32_t c0 = 0; c0 < gs; c0 += PSELL_LANES) {
            const uint32_t nrow = min(gs - c0, (uint32_t)PSELL_LANES);
            uint32_t l = 0, row = 0;
            if (lane < nrow) {
                if (c0 < GL) {
                    l = g_len[c0 + lane];
                    row = g_row[c0 + lane];
                } else {
                    const uint32_t q = P.gbuf[p0 + c0 + lane];
                    l = P.plen[q];
                    row = P.prow[q];
                }
            }
            const uint32_t total = wave_sum_u32(l), longest = wave_max_u32(l);
            const double dense_bytes = 256.0 * (double)(nu + 1 + (uint32_t)P.ks_rows);
            const double masked_bytes = 256.0 * (double)(longest + (narrow ? 1u : 2u) + (uint32_t)P.ks_rows);
            const double budget = 8.0 * (double)total + 4.0 * (double)nrow;
            const double cost = fmin(dense_bytes, masked_bytes);
            bool worth = cost <= budget;
            if (!worth && ((P.pass_w == 1 && (cost <= relax * budget || longest > (uint32_t)PSELL_MIXED_NARROW_MAX)) || (P.pass_w == 0 && cost <= relax0 * budget))) {
                const double over = cost - budget;
                if (over <= allowance) {
                    allowance -= over;
                    worth = true;
                } else if (P.pass_w == 1 && longest > (uint32_t)PSELL_MIXED_NARROW_MAX && over <= wide_allowance) {
                    wide_allowance -= over;
                    worth = true;
                }
            }
            if (worth) {
                const bool masked = masked_bytes < (1.0 - mask_gain) * dense_bytes;
                const uint32_t dest = narrow ? (masked ? 1u : 0u) : (masked ? 3u : 2u);
                if (lane < nrow) {
                    P.rec_row[p0 + emitted + lane] = row;
                    P.rec_meta[p0 + emitted + lane] = (uint8_t)(dest | ((lane + 1 == nrow ? 1u : 0u) << 3) | ((masked ? 2u : 1u) << 4));
                    P.rec_gid[p0 + emitted + lane] = pid;
                }
                any = true;
            } else if (lane < nrow) {
                P.rec_row[p0 + emitted + lane] = row;
                P.rec_meta[p0 + emitted + lane] = 4;
                P.rec_gid[p0 + emitted + lane] = 0;
            }
            emitted += nrow;
        }
        if (any) {
            if (lane < nu) P.pcols[pat_base + pat_off + lane] = u_reg;
            if (lane == 0) {
                P.psize[p0 + pid] = nu;
                P.poff[p0 + pid] = pat_off;
            }
            pat_off += nu;
            ++npat;
        }
        gs = 0;
        nu = 0;
        u_reg = 0xffffffffu;
        WAVE_LDS_SYNC();
    };
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t cnt = pass == 0 ? p1 - p0 : nd;
        uint32_t ndn = 0;
        // (a block of 64 rows' positions, lengths and offsets sits in the lanes.  The union and the row are compared through
        // v_readlane with a uniform lane index -- no LDS round trip per element.)
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t nb = min(64u, cnt - base);
            uint32_t myq = 0, mylen = 0, mybeg_lo = 0, mybeg_hi = 0, myrow = 0;
            if (lane < nb) {
                myq = pass == 0 ? p0 + base + lane : P.dbuf[p0 + base + lane];
                mylen = P.plen[myq];
                myrow = P.prow[myq];
                const uint64_t bg = P.pbeg[myq];
                mybeg_lo = (uint32_t)bg;
                mybeg_hi = (uint32_t)(bg >> 32);
            }
            uint32_t c_next;  // (the NEXT row's transcripts are on their way while the current row is merged)
            {
                const uint32_t len0 = rdlane(mylen, 0);
                const uint64_t beg0 = ((uint64_t)rdlane(mybeg_hi, 0) << 32) | rdlane(mybeg_lo, 0);
                c_next = lane < len0 ? P.col[beg0 + lane] : 0xffffffffu;
            }
            for (uint32_t t = 0; t < nb; ++t) {
                const uint32_t q = rdlane(myq, t);
                const uint32_t len = rdlane(mylen, t);
                const uint32_t row = rdlane(myrow, t);
                const uint32_t c = c_next;  // lane i < len: the row's i-th transcript
                if (t + 1 < nb) {
                    const uint32_t len2 = rdlane(mylen, t + 1);
                    const uint64_t beg2 = ((uint64_t)rdlane(mybeg_hi, t + 1) << 32) | rdlane(mybeg_lo, t + 1);
                    c_next = lane < len2 ? P.col[beg2 + lane] : 0xffffffffu;
                }
                bool found = false;
                uint32_t rank = 0;  // transcripts of the union below this lane's
                for (uint32_t j = 0; j < nu; ++j) {
                    const uint32_t u = rdlane(u_reg, j);
                    found |= u == c;
                    rank += u < c;
                }
                const bool isnew = lane < len && !found;
                const uint64_t bal = __ballot(isnew);
                const uint32_t nnew = (uint32_t)__popcll(bal), nt = nu + nnew;
                if (nt <= P.cap && (gs < (uint32_t)PSELL_LANES || (nt + 3) / 4 == (nu + 3) / 4) && gs < S2_MAX_GROUP) {
                    if (nnew) {
                        uint32_t before = 0;  // new transcripts below this lane's member of the union
                        for (uint64_t rest = bal; rest; rest &= rest - 1) {
                            const uint32_t i = (uint32_t)__builtin_ctzll(rest);
                            before += rdlane(c, i) < u_reg;
                        }
                        if (lane < nu) smerge[lane + before] = u_reg;
                        if (isnew) smerge[rank + (uint32_t)__popcll(bal & lanes_below())] = c;
                        WAVE_LDS_SYNC();
                        u_reg = lane < nt ? smerge[lane] : 0xffffffffu;
                        WAVE_LDS_SYNC();
                        nu = nt;
                    }
                    if (lane == 0) {
                        if (gs < GL) {
                            g_row[gs] = row;
                            g_len[gs] = len;
                        } else {
                            P.gbuf[p0 + gs] = q;
                        }
                    }
                    ++gs;
                    continue;
                }
                if (pass == 0 && gs < 48 && nu + 1 < P.cap) {
                    if (lane == 0) P.dbuf[p0 + ndn] = q;  // an outlier (a neighbouring gene's isoform): second pass
                    ++ndn;
                    continue;
                }
                close_group();
                u_reg = lane < len ? c : 0xffffffffu;
                nu = len;
                if (lane == 0) {
                    g_row[0] = row;
                    g_len[0] = len;
                }
                gs = 1;
            }
        }
        close_group();
        if (pass == 0) nd = ndn;
        __syncthreads();
    }
    if (lane == 0) P.chunk_npats[blockIdx.x] = npat;
}

// the records of a pass into the four lists and the rows left over (each in record order: chunk by chunk, group by group)
__global__ void s2_recflag_kernel(uint32_t N, const uint8_t *meta, uint32_t *f0, uint32_t *f1, uint32_t *f2, uint32_t *f3, uint32_t *f4)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t d = meta[i] & 7u;
    f0[i] = d == 0; f1[i] = d == 1; f2[i] = d == 2; f3[i] = d == 3; f4[i] = d == 4;
}
struct ListPtrs {
    uint32_t *rows[4], *ends[4], *gid[4];
    uint8_t *form[4];
    uint32_t *left;
};
__global__ void s2_recscatter_kernel(uint32_t N, const uint32_t *rec_row, const uint8_t *meta, const uint32_t *rec_gid, const uint32_t *gbase,
                                     const uint32_t *p0, const uint32_t *p1, const uint32_t *p2, const uint32_t *p3, const uint32_t *p4, ListPtrs L)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t d = meta[i] & 7u;
    if (d == 4) {
        L.left[p4[i]] = rec_row[i];
        return;
    }
    const uint32_t pos = d == 0 ? p0[i] : (d == 1 ? p1[i] : (d == 2 ? p2[i] : p3[i]));
    L.rows[d][pos] = rec_row[i];
    L.ends[d][pos] = (meta[i] >> 3) & 1u;
    L.form[d][pos] = (uint8_t)(meta[i] >> 4);
    L.gid[d][pos] = rec_gid[i] + gbase[i / S2_CHUNK];
}
// the patterns of a pass, chunk by chunk: sizes, then the transcript ids
__global__ void s2_patflag_kernel(uint32_t N, const uint32_t *chunk_npats, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flag[i] = (i % S2_CHUNK) < chunk_npats[i / S2_CHUNK];
}
__global__ void s2_patsize_kernel(uint32_t N, const uint32_t *flag, const uint32_t *pos, const uint32_t *psize, uint32_t *sizes, uint32_t *src)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && flag[i]) {
        sizes[pos[i]] = psize[i];
        src[pos[i]] = i;
    }
}
__global__ void s2_patcopy_kernel(uint32_t npat, const uint32_t *src, const uint32_t *sizes, const uint32_t *ptr, const uint32_t *poff,
                                  const uint64_t *plps, const uint32_t *pcols, uint32_t *pat_col, uint32_t col_base)
{
    const uint32_t k = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (k >= npat) return;
    const uint32_t i = src[k], t = threadIdx.x & 31u;
    const uint32_t p0 = (i / S2_CHUNK) * S2_CHUNK;
    if (t < sizes[k]) pat_col[col_base + ptr[k] + t] = pcols[plps[p0] + poff[i] + t];
}
__global__ void s2_blockkey_kernel(PsellDevIn X, uint32_t N, const uint32_t *rows, uint32_t block_rows, uint32_t *key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t len = (uint32_t)(X.rowptr[rows[i] + 1] - X.rowptr[rows[i]]);
    key[i] = ((i / block_rows) << 11) | (2047u - len);  // (len <= 1024): blocks in order, longest rows first, ties in order
}
__global__ void s2_fill_u32_kernel(uint32_t N, uint32_t *p, uint32_t v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) p[i] = v;
}

struct DevClock {  // POLEE_BUILD_TIMING=1: phases of the device builder on stderr (each mark waits for the stream)
    bool on = getenv("POLEE_BUILD_TIMING") != nullptr;
    hipStream_t stream;
    double t0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    explicit DevClock(hipStream_t s) : stream(s), t0(now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        (void)hipStreamSynchronize(stream);
        fprintf(stderr, "[psell device build]   . %-26s %.4f s\n", what, now() - t0);
        t0 = now();
    }
};

struct PackLists {
    DevBuf<uint32_t> rows[4], ends[4], gid[4], left;
    DevBuf<uint8_t> form[4];
    uint32_t count[4] = {0, 0, 0, 0}, nleft = 0;
    DevBuf<uint32_t> pat_sizes, pat_src, pat_ptr, poff, pcols;
    DevBuf<uint64_t> plps;
    uint32_t npat = 0, pat_cols = 0;
};

polee_status sort_by_first_col(polee_ctx *ctx, Scratch &tmp, const PsellDevIn &X, const uint32_t *rows, uint32_t N, DevBuf<uint32_t> &out)
{
    hipStream_t stream = ctx->stream;
    POLEE_TRY(out.alloc(ctx, (size_t)N + 1));
    if (N == 0) return POLEE_OK;
    DevBuf<uint32_t> key, key_s;
    POLEE_TRY(key.alloc(ctx, N));
    POLEE_TRY(key_s.alloc(ctx, N));
    hipLaunchKernelGGL(s2_firstcol_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, X, N, rows, key.p);
    POLEE_KERNEL_CHECK(ctx);
    size_t bytes = 0;
    PD_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key_s.p, rows, out.p, (size_t)N, 0, 32, stream));
    PD_HIP(tmp.need(bytes));
    PD_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key_s.p, rows, out.p, (size_t)N, 0, 32, stream));
    PD_HIP(hipStreamSynchronize(stream));
    return POLEE_OK;
}

// out = the rows whose class (masked) equals k, in order
polee_status select_class(polee_ctx *ctx, Scratch &tmp, const uint32_t *rows, const uint8_t *cls, uint32_t N, uint8_t k, uint32_t mask,
                          DevBuf<uint32_t> &flag, DevBuf<uint32_t> &pos, DevBuf<uint32_t> &out, uint32_t &count)
{
    hipStream_t stream = ctx->stream;
    count = 0;
    POLEE_TRY(out.alloc(ctx, 1));
    if (N == 0) return POLEE_OK;
    POLEE_TRY(flag.alloc(ctx, (size_t)N + 1));
    POLEE_TRY(pos.alloc(ctx, (size_t)N + 1));
    PD_HIP(hipMemsetAsync(flag.p + N, 0, 4, stream));
    hipLaunchKernelGGL(s2_flag_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, cls, k, mask, flag.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, flag.p, pos.p, 0u, (size_t)N + 1, stream));
    PD_HIP(hipMemcpyAsync(&count, pos.p + N, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    POLEE_TRY(out.alloc(ctx, (size_t)count + 1));
    hipLaunchKernelGGL(s2_scatter_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, flag.p, pos.p, rows, out.p);
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

polee_status concat2(polee_ctx *ctx, const DevBuf<uint32_t> &a, uint32_t na, const DevBuf<uint32_t> &b, uint32_t nb, DevBuf<uint32_t> &out)
{
    POLEE_TRY(out.alloc(ctx, (size_t)na + nb + 1));
    if (na) PD_HIP(hipMemcpyAsync(out.p, a.p, (size_t)na * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (nb) PD_HIP(hipMemcpyAsync(out.p + na, b.p, (size_t)nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return POLEE_OK;
}

// one pass of the packing over `pool` (already in first-transcript order)
polee_status run_pack(polee_ctx *ctx, Scratch &tmp, const PsellDevIn &X, uint64_t nnz_total, const DevBuf<uint32_t> &pool, uint32_t Np, uint32_t cap,
                      int pass_w, uint32_t gid_base, PackLists &L)
{
    hipStream_t stream = ctx->stream;
    for (int d = 0; d < 4; ++d) L.count[d] = 0;
    L.nleft = 0;
    L.npat = 0;
    L.pat_cols = 0;
    if (Np == 0) return POLEE_OK;
    const unsigned TB = 256;
    const uint32_t nch = (Np + S2_CHUNK - 1) / S2_CHUNK;
    DevBuf<uint32_t> plen, rec_row, rec_gid, chunk_npats, gbase, psize, gbuf, dbuf, f[5], p[5], pflag, ppos;
    DevBuf<uint64_t> pbeg;
    DevBuf<uint8_t> rec_meta;
    POLEE_TRY(plen.alloc(ctx, (size_t)Np + 1));
    POLEE_TRY(pbeg.alloc(ctx, (size_t)Np + 1));
    POLEE_TRY(L.plps.alloc(ctx, (size_t)Np + 1));
    PD_HIP(hipMemsetAsync(plen.p + Np, 0, 4, stream));
    hipLaunchKernelGGL(s2_poolinfo_kernel, dim3((Np + TB - 1) / TB), dim3(TB), 0, stream, X, Np, pool.p, plen.p, pbeg.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(plen.p, ToU64()), L.plps.p, (uint64_t)0, (size_t)Np + 1, stream));
    uint64_t pool_nnz = 0;
    PD_HIP(hipMemcpyAsync(&pool_nnz, L.plps.p + Np, 8, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (pool_nnz >= (1ull << 32)) return fail(ctx, POLEE_ERR_UNSUPPORTED, "device layout build: more than 2^32 non-zeros in leftover rows");
    const double over_budget = 0.02;
    const double pool_csr_bytes = 8.0 * (double)pool_nnz + 4.0 * (double)Np;
    const double matrix_csr_bytes = 8.0 * (double)nnz_total + 4.0 * (double)X.m;
    S2Pack P;
    P.col = X.col; P.prow = pool.p; P.plen = plen.p; P.pbeg = pbeg.p; P.plps = L.plps.p; P.Np = Np; P.cap = cap; P.pass_w = pass_w;
    P.ks_rows = X.ks ? 1 : 0;
    P.factor = over_budget * matrix_csr_bytes / std::max(pool_csr_bytes, 1.0);
    POLEE_TRY(rec_row.alloc(ctx, Np));
    POLEE_TRY(rec_gid.alloc(ctx, Np));
    POLEE_TRY(rec_meta.alloc(ctx, Np));
    POLEE_TRY(chunk_npats.alloc(ctx, (size_t)nch + 1));
    POLEE_TRY(gbase.alloc(ctx, (size_t)nch + 1));
    POLEE_TRY(psize.alloc(ctx, Np));
    POLEE_TRY(L.poff.alloc(ctx, Np));
    POLEE_TRY(L.pcols.alloc(ctx, (size_t)pool_nnz + 1));
    POLEE_TRY(gbuf.alloc(ctx, Np));
    POLEE_TRY(dbuf.alloc(ctx, Np));
    P.rec_row = rec_row.p; P.rec_gid = rec_gid.p; P.rec_meta = rec_meta.p; P.chunk_npats = chunk_npats.p; P.psize = psize.p; P.poff = L.poff.p;
    P.pcols = L.pcols.p; P.gbuf = gbuf.p; P.dbuf = dbuf.p;
    PD_HIP(hipMemsetAsync(chunk_npats.p + nch, 0, 4, stream));
    DevClock clk(stream);
    clk.lap("pack: pool info, buffers");
    hipLaunchKernelGGL(s2_pack_kernel, dim3(nch), dim3(64), 0, stream, P);
    POLEE_KERNEL_CHECK(ctx);
    clk.lap("pack: greedy walk");
    PD_HIP(exclusive_sum(tmp, chunk_npats.p, gbase.p, gid_base, (size_t)nch + 1, stream));
    // the four lists + the rows left
    uint32_t cnt[5];
    for (int d = 0; d < 5; ++d) {
        POLEE_TRY(f[d].alloc(ctx, (size_t)Np + 1));
        POLEE_TRY(p[d].alloc(ctx, (size_t)Np + 1));
        PD_HIP(hipMemsetAsync(f[d].p + Np, 0, 4, stream));
    }
    hipLaunchKernelGGL(s2_recflag_kernel, dim3((Np + TB - 1) / TB), dim3(TB), 0, stream, Np, rec_meta.p, f[0].p, f[1].p, f[2].p, f[3].p, f[4].p);
    POLEE_KERNEL_CHECK(ctx);
    for (int d = 0; d < 5; ++d) {
        PD_HIP(exclusive_sum(tmp, f[d].p, p[d].p, 0u, (size_t)Np + 1, stream));
        PD_HIP(hipMemcpyAsync(&cnt[d], p[d].p + Np, 4, hipMemcpyDeviceToHost, stream));
    }
    uint32_t gend = 0;
    PD_HIP(hipMemcpyAsync(&gend, gbase.p + nch, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    L.npat = gend - gid_base;
    ListPtrs LP;
    for (int d = 0; d < 4; ++d) {
        L.count[d] = cnt[d];
        POLEE_TRY(L.rows[d].alloc(ctx, (size_t)cnt[d] + 1));
        POLEE_TRY(L.ends[d].alloc(ctx, (size_t)cnt[d] + 1));
        POLEE_TRY(L.gid[d].alloc(ctx, (size_t)cnt[d] + 1));
        POLEE_TRY(L.form[d].alloc(ctx, (size_t)cnt[d] + 1));
        LP.rows[d] = L.rows[d].p; LP.ends[d] = L.ends[d].p; LP.gid[d] = L.gid[d].p; LP.form[d] = L.form[d].p;
    }
    L.nleft = cnt[4];
    POLEE_TRY(L.left.alloc(ctx, (size_t)cnt[4] + 1));
    LP.left = L.left.p;
    hipLaunchKernelGGL(s2_recscatter_kernel, dim3((Np + TB - 1) / TB), dim3(TB), 0, stream, Np, rec_row.p, rec_meta.p, rec_gid.p, gbase.p, p[0].p, p[1].p,
                       p[2].p, p[3].p, p[4].p, LP);
    POLEE_KERNEL_CHECK(ctx);
    // the patterns: sizes in (chunk, group) order, offsets, where each one's ids lie
    POLEE_TRY(pflag.alloc(ctx, (size_t)Np + 1));
    POLEE_TRY(ppos.alloc(ctx, (size_t)Np + 1));
    PD_HIP(hipMemsetAsync(pflag.p + Np, 0, 4, stream));
    hipLaunchKernelGGL(s2_patflag_kernel, dim3((Np + TB - 1) / TB), dim3(TB), 0, stream, Np, chunk_npats.p, pflag.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, pflag.p, ppos.p, 0u, (size_t)Np + 1, stream));
    POLEE_TRY(L.pat_sizes.alloc(ctx, (size_t)L.npat + 1));
    POLEE_TRY(L.pat_src.alloc(ctx, (size_t)L.npat + 1));
    POLEE_TRY(L.pat_ptr.alloc(ctx, (size_t)L.npat + 1));
    PD_HIP(hipMemsetAsync(L.pat_sizes.p + L.npat, 0, 4, stream));
    hipLaunchKernelGGL(s2_patsize_kernel, dim3((Np + TB - 1) / TB), dim3(TB), 0, stream, Np, pflag.p, ppos.p, psize.p, L.pat_sizes.p, L.pat_src.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, L.pat_sizes.p, L.pat_ptr.p, 0u, (size_t)L.npat + 1, stream));
    PD_HIP(hipMemcpyAsync(&L.pat_cols, L.pat_ptr.p + L.npat, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    clk.lap("pack: lists and patterns");
    return POLEE_OK;
}

}  // namespace

polee_status psell_device_stage2(polee_ctx *ctx, const PsellDevIn &X, PsellDevRuns &R, PsellHost &out, PsellDevRowsOwned &W, bool &needs_host)
{
    hipStream_t stream = ctx->stream;
    needs_host = false;
    Scratch tmp(ctx);
    const unsigned TB = 256;
    const uint64_t nnz_total = (uint64_t)out.nnz;
    // candidates in the order of their first transcript; by length: <= 16 first pass, 17..32 second pass, longer: mixed streams
    DevBuf<uint32_t> cand, flag, pos, pool0, l1, l2;
    DevBuf<uint8_t> cls;
    const uint32_t nrb = (uint32_t)R.n_rb;
    POLEE_TRY(sort_by_first_col(ctx, tmp, X, R.rb.p, nrb, cand));
    uint32_t n0 = 0, n1 = 0, n2 = 0;
    POLEE_TRY(cls.alloc(ctx, (size_t)nrb + 1));
    if (nrb) {
        hipLaunchKernelGGL(s2_lenclass_kernel, dim3((nrb + TB - 1) / TB), dim3(TB), 0, stream, X, nrb, cand.p, (uint32_t)PSELL_NARROW_MAX, (uint32_t)PSELL_WIDE_MAX, cls.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    POLEE_TRY(select_class(ctx, tmp, cand.p, cls.p, nrb, 0, 0xffu, flag, pos, pool0, n0));
    POLEE_TRY(select_class(ctx, tmp, cand.p, cls.p, nrb, 1, 0xffu, flag, pos, l1, n1));
    POLEE_TRY(select_class(ctx, tmp, cand.p, cls.p, nrb, 2, 0xffu, flag, pos, l2, n2));
    DevClock clk(stream);
    clk.lap("candidates sorted, split");
    PackLists P0, P1;
    POLEE_TRY(run_pack(ctx, tmp, X, nnz_total, pool0, n0, (uint32_t)PSELL_NARROW_MAX, 0, 0u, P0));
    DevBuf<uint32_t> np_unsorted, pool1;
    POLEE_TRY(concat2(ctx, l1, n1, P0.left, P0.nleft, np_unsorted));
    const uint32_t np1 = n1 + P0.nleft;
    POLEE_TRY(sort_by_first_col(ctx, tmp, X, np_unsorted.p, np1, pool1));
    clk.lap("(first pass)");
    POLEE_TRY(run_pack(ctx, tmp, X, nnz_total, pool1, np1, (uint32_t)PSELL_WIDE_MAX, 1, P0.npat, P1));
    clk.lap("(second pass)");
    // the mixed streams: what found no company, in first-transcript order; BN: rows of <= 15 transcripts, B: the others; inside every
    // block of 1024 rows by descending length
    DevBuf<uint32_t> kb_unsorted, kb, rbn, wide;
    POLEE_TRY(concat2(ctx, l2, n2, P1.left, P1.nleft, kb_unsorted));
    const uint32_t nkb = n2 + P1.nleft;
    POLEE_TRY(sort_by_first_col(ctx, tmp, X, kb_unsorted.p, nkb, kb));
    uint32_t nbn = 0, nwide = 0;
    DevBuf<uint8_t> cls2;
    POLEE_TRY(cls2.alloc(ctx, (size_t)nkb + 1));
    if (nkb) {
        hipLaunchKernelGGL(s2_lenclass_kernel, dim3((nkb + TB - 1) / TB), dim3(TB), 0, stream, X, nkb, kb.p, (uint32_t)PSELL_MIXED_NARROW_MAX, 0xffffffffu, cls2.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    POLEE_TRY(select_class(ctx, tmp, kb.p, cls2.p, nkb, 0, 0xffu, flag, pos, rbn, nbn));
    POLEE_TRY(select_class(ctx, tmp, kb.p, cls2.p, nkb, 1, 0xffu, flag, pos, wide, nwide));
    DevBuf<uint32_t> rbn_s, wide_s;
    uint64_t mixed_nnz = 0;
    for (int which = 0; which < 2; ++which) {
        DevBuf<uint32_t> &src = which == 0 ? rbn : wide, &dst = which == 0 ? rbn_s : wide_s;
        const uint32_t N = which == 0 ? nbn : nwide;
        POLEE_TRY(dst.alloc(ctx, (size_t)N + 1));
        if (N == 0) continue;
        DevBuf<uint32_t> key, key_s, lens;
        DevBuf<uint64_t> lsum;
        POLEE_TRY(key.alloc(ctx, N));
        POLEE_TRY(key_s.alloc(ctx, N));
        hipLaunchKernelGGL(s2_blockkey_kernel, dim3((N + TB - 1) / TB), dim3(TB), 0, stream, X, N, src.p, (uint32_t)(PSELL_LANES * PSELL_TILE_SLICES_B), key.p);
        POLEE_KERNEL_CHECK(ctx);
        size_t bytes = 0;
        PD_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key_s.p, src.p, dst.p, (size_t)N, 0, 32, stream));
        PD_HIP(tmp.need(bytes));
        PD_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key_s.p, src.p, dst.p, (size_t)N, 0, 32, stream));
        // their non-zeros (for the CSR question below): 2047 - (key & 2047) summed
        std::vector<uint32_t> hk(N);
        PD_HIP(hipMemcpyAsync(hk.data(), key.p, (size_t)N * 4, hipMemcpyDeviceToHost, stream));
        PD_HIP(hipStreamSynchronize(stream));
        for (uint32_t v : hk) mixed_nnz += 2047u - (v & 2047u);
    }
    // Stream C (rows kept in CSR): the host builder simulates the mixed streams' tiles row by row and keeps rows in CSR only when
    // they are more than a tenth of the matrix -- which they cannot be when ALL the mixed rows are less.  Otherwise (a matrix
    // without structure) the layout is the host builder's to make.
    if ((double)mixed_nnz >= 0.10 * (double)nnz_total && mixed_nnz > 0) {
        needs_host = true;
        return POLEE_OK;
    }
    clk.lap("mixed streams");
    // ---- the ordered rows of the sliced streams: A1 = exact runs, first-pass unions, second-pass narrow unions; A1M; A2; A2M; BN; B
    const uint32_t nA1 = (uint32_t)R.n_a1 + P0.count[0] + P1.count[0], nA1M = P0.count[1] + P1.count[1];
    const uint32_t nA2 = (uint32_t)R.n_a2 + P0.count[2] + P1.count[2], nA2M = P0.count[3] + P1.count[3];
    const uint64_t Nr64 = (uint64_t)nA1 + nA1M + nA2 + nA2M + nbn + nwide;
    if (Nr64 >= (1ull << 32) - 1) return fail(ctx, POLEE_ERR_UNSUPPORTED, "device layout build: too many rows");
    const uint32_t Nr = (uint32_t)Nr64;
    W.Nr = Nr;
    W.bounds[0] = 0; W.bounds[1] = nA1; W.bounds[2] = W.bounds[1] + nA1M; W.bounds[3] = W.bounds[2] + nA2; W.bounds[4] = W.bounds[3] + nA2M;
    W.bounds[5] = W.bounds[4] + nbn; W.bounds[6] = Nr;
    POLEE_TRY(W.rows.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(W.run_end.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(W.gid.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(W.form.alloc(ctx, (size_t)Nr + 1));
    PD_HIP(hipMemsetAsync(W.run_end.p, 0, ((size_t)Nr + 1) * 4, stream));
    PD_HIP(hipMemsetAsync(W.gid.p, 0, ((size_t)Nr + 1) * 4, stream));
    PD_HIP(hipMemsetAsync(W.form.p, 0, (size_t)Nr + 1, stream));
    size_t at = 0;
    auto put_u32 = [&](uint32_t *dst, const uint32_t *src, size_t cnt) -> polee_status {
        if (cnt) PD_HIP(hipMemcpyAsync(dst + at, src, cnt * 4, hipMemcpyDeviceToDevice, stream));
        return POLEE_OK;
    };
    auto put_list = [&](const PackLists &L, int d) -> polee_status {
        const size_t cnt = L.count[d];
        POLEE_TRY(put_u32(W.rows.p, L.rows[d].p, cnt));
        POLEE_TRY(put_u32(W.run_end.p, L.ends[d].p, cnt));
        POLEE_TRY(put_u32(W.gid.p, L.gid[d].p, cnt));
        if (cnt) PD_HIP(hipMemcpyAsync(W.form.p + at, L.form[d].p, cnt, hipMemcpyDeviceToDevice, stream));
        at += cnt;
        return POLEE_OK;
    };
    POLEE_TRY(put_u32(W.rows.p, R.a1_rows.p, R.n_a1));
    POLEE_TRY(put_u32(W.run_end.p, R.a1_ends.p, R.n_a1));
    at += R.n_a1;
    POLEE_TRY(put_list(P0, 0));
    POLEE_TRY(put_list(P1, 0));
    POLEE_TRY(put_list(P0, 1));
    POLEE_TRY(put_list(P1, 1));
    POLEE_TRY(put_u32(W.rows.p, R.a2_rows.p, R.n_a2));
    POLEE_TRY(put_u32(W.run_end.p, R.a2_ends.p, R.n_a2));
    at += R.n_a2;
    POLEE_TRY(put_list(P0, 2));
    POLEE_TRY(put_list(P1, 2));
    POLEE_TRY(put_list(P0, 3));
    POLEE_TRY(put_list(P1, 3));
    POLEE_TRY(put_u32(W.rows.p, rbn_s.p, nbn));
    at += nbn;
    POLEE_TRY(put_u32(W.rows.p, wide_s.p, nwide));
    at += nwide;
    // the groups' transcript sets: first pass's, then second pass's
    W.npat = (size_t)P0.npat + P1.npat;
    const uint32_t ncols = P0.pat_cols + P1.pat_cols;
    POLEE_TRY(W.pat_ptr.alloc(ctx, W.npat + 1));
    POLEE_TRY(W.pat_col.alloc(ctx, (size_t)ncols + 1));
    {
        // pat_ptr = [P0.ptr ..., P0.cols + P1.ptr ...]
        if (P0.npat) PD_HIP(hipMemcpyAsync(W.pat_ptr.p, P0.pat_ptr.p, (size_t)P0.npat * 4, hipMemcpyDeviceToDevice, stream));
        std::vector<uint32_t> h1((size_t)P1.npat + 1, 0);
        if (P1.npat) PD_HIP(hipMemcpyAsync(h1.data(), P1.pat_ptr.p, ((size_t)P1.npat + 1) * 4, hipMemcpyDeviceToHost, stream));
        PD_HIP(hipStreamSynchronize(stream));
        for (auto &v : h1) v += P0.pat_cols;
        if (P1.npat == 0) h1[0] = P0.pat_cols;
        PD_HIP(hipMemcpyAsync(W.pat_ptr.p + P0.npat, h1.data(), ((size_t)P1.npat + 1) * 4, hipMemcpyHostToDevice, stream));
        PD_HIP(hipStreamSynchronize(stream));
        if (P0.npat) {
            hipLaunchKernelGGL(s2_patcopy_kernel, dim3((P0.npat + 7) / 8), dim3(256), 0, stream, P0.npat, P0.pat_src.p, P0.pat_sizes.p, P0.pat_ptr.p,
                               P0.poff.p, P0.plps.p, P0.pcols.p, W.pat_col.p, 0u);
            POLEE_KERNEL_CHECK(ctx);
        }
        if (P1.npat) {
            hipLaunchKernelGGL(s2_patcopy_kernel, dim3((P1.npat + 7) / 8), dim3(256), 0, stream, P1.npat, P1.pat_src.p, P1.pat_sizes.p, P1.pat_ptr.p,
                               P1.poff.p, P1.plps.p, P1.pcols.p, W.pat_col.p, P0.pat_cols);
            POLEE_KERNEL_CHECK(ctx);
        }
    }
    PD_HIP(hipStreamSynchronize(stream));
    clk.lap("ordered rows assembled");
    return POLEE_OK;
}

polee_status psell_device_rows_to_host(polee_ctx *ctx, const PsellDevRowsOwned &W, PsellHost &out, PsellRows &H)
{
    H = PsellRows();
    const size_t nu = W.bounds[4];  // rows of the uniform streams: run_end / form / gid cover these
    H.rows.resize(W.Nr);
    H.run_end.resize(nu);
    H.row_gid.resize(nu);
    H.row_form.resize(nu);
    H.pat_ptr.resize(W.npat + 1);
    POLEE_TRY(W.rows.download(ctx, H.rows.data(), W.Nr));
    POLEE_TRY(W.run_end.download(ctx, H.run_end.data(), nu));
    POLEE_TRY(W.gid.download(ctx, H.row_gid.data(), nu));
    POLEE_TRY(W.form.download(ctx, H.row_form.data(), nu));
    POLEE_TRY(W.pat_ptr.download(ctx, H.pat_ptr.data(), W.npat + 1));
    H.pat_col.resize(H.pat_ptr.back());
    POLEE_TRY(W.pat_col.download(ctx, H.pat_col.data(), H.pat_col.size()));
    out.rows_a1 = (int64_t)W.bounds[1];
    out.rows_a1m = (int64_t)W.bounds[2];
    out.rows_a2 = (int64_t)W.bounds[3];
    out.rows_a = (int64_t)W.bounds[4];
    out.rows_s = (int64_t)W.bounds[5];
    return POLEE_OK;
}

polee_status psell_device_stage3(polee_ctx *ctx, const PsellDevIn &X, const PsellDevRows &W, PsellHost &out, PsellDevOut &D,
                                 bool want_debug)
{
    hipStream_t stream = ctx->stream;
    const uint32_t Nr = (uint32_t)W.Nr;
    out.rows_a1 = (int64_t)W.bounds[1];
    out.rows_a1m = (int64_t)W.bounds[2];
    out.rows_a2 = (int64_t)W.bounds[3];
    out.rows_a = (int64_t)W.bounds[4];
    out.rows_s = (int64_t)W.bounds[5];
    out.slice_off.assign(1, 0);
    out.tile_slice.assign(1, 0);
    out.tile_dict.assign(1, 0);
    if (Nr == 0) {
        POLEE_TRY(D.data.alloc(ctx, 2048));
        PD_HIP(hipMemsetAsync(D.data.p, 0, 2048, stream));
        D.data_bytes = 0;
        return POLEE_OK;
    }
    S3In A;
    A.rowptr = X.rowptr; A.col = X.col; A.val = X.val; A.ks = X.ks;
    A.rows = W.rows; A.run_end = W.run_end; A.gid = W.gid; A.form = W.form; A.pat_ptr = W.pat_ptr; A.pat_col = W.pat_col;
    A.Nr = Nr;
    A.n = (uint32_t)X.n;
    for (int q = 0; q < 7; ++q) A.bounds[q] = (uint32_t)W.bounds[q];
    static const size_t seg_env = getenv("POLEE_PSELL_SEG_ROWS") ? (size_t)atoll(getenv("POLEE_PSELL_SEG_ROWS")) : 0;  // (tests)
    const uint32_t seg_rows = (uint32_t)(seg_env >= 64 ? seg_env : (size_t)1 << 18);
    const uint32_t max_segs = Nr / seg_rows + (uint32_t)(W.bounds[6] - W.bounds[4]) / std::min<uint32_t>(seg_rows, PSELL_MIXED_SEG_ROWS) + 8;
    Scratch tmp(ctx);
    DevBuf<uint32_t> rlen, head, endflag, hscan, escan, st_start, endpos, nseg_d;
    DevBuf<uint64_t> lenps, totals;
    DevBuf<SegDesc> segs;
    DevBuf<SegAux> aux;
    POLEE_TRY(rlen.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(head.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(endflag.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(hscan.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(escan.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(lenps.alloc(ctx, (size_t)Nr + 1));
    POLEE_TRY(segs.alloc(ctx, max_segs));
    POLEE_TRY(aux.alloc(ctx, max_segs));
    POLEE_TRY(nseg_d.alloc(ctx, 2));
    POLEE_TRY(totals.alloc(ctx, 2));
    PD_HIP(hipMemsetAsync(rlen.p + Nr, 0, 4, stream));
    PD_HIP(hipMemsetAsync(head.p + Nr, 0, 4, stream));
    PD_HIP(hipMemsetAsync(endflag.p + Nr, 0, 4, stream));
    PD_HIP(hipMemsetAsync(nseg_d.p, 0, 8, stream));
    const unsigned TB = 256;
    hipLaunchKernelGGL(s3_rowinfo_kernel, dim3((Nr + TB - 1) / TB), dim3(TB), 0, stream, A, rlen.p, head.p, endflag.p);
    POLEE_KERNEL_CHECK(ctx);
    hipLaunchKernelGGL(s3_segments_kernel, dim3(1), dim3(1), 0, stream, A, seg_rows, segs.p, nseg_d.p, head.p, max_segs);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(exclusive_sum(tmp, head.p, hscan.p, 0u, (size_t)Nr + 1, stream));
    PD_HIP(exclusive_sum(tmp, endflag.p, escan.p, 0u, (size_t)Nr + 1, stream));
    PD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(rlen.p, ToU64()), lenps.p, (uint64_t)0, (size_t)Nr + 1, stream));
    uint32_t nseg = 0, nstretch = 0, nends = 0;
    PD_HIP(hipMemcpyAsync(&nseg, nseg_d.p, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(&nstretch, hscan.p + Nr, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(&nends, escan.p + Nr, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (nseg > max_segs) return fail(ctx, POLEE_ERR_HIP, "device layout build: %u segments, room for %u", nseg, max_segs);
    POLEE_TRY(st_start.alloc(ctx, (size_t)nstretch + 1));
    POLEE_TRY(endpos.alloc(ctx, (size_t)nends + 1));
    hipLaunchKernelGGL(s3_scatter_kernel, dim3((Nr + 1 + TB - 1) / TB), dim3(TB), 0, stream, Nr, head.p, hscan.p, endflag.p, escan.p,
                       st_start.p, endpos.p);
    POLEE_KERNEL_CHECK(ctx);
    hipLaunchKernelGGL(s3_segaux_kernel, dim3(1), dim3(1), 0, stream, A, segs.p, nseg, hscan.p, escan.p, lenps.p, aux.p, totals.p);
    POLEE_KERNEL_CHECK(ctx);
    std::vector<SegDesc> h_segs(nseg);
    std::vector<SegAux> h_aux(nseg);
    uint64_t h_tot[2] = {0, 0};
    PD_HIP(hipMemcpyAsync(h_segs.data(), segs.p, sizeof(SegDesc) * nseg, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(h_aux.data(), aux.p, sizeof(SegAux) * nseg, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(h_tot, totals.p, 16, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (h_tot[0] >= (1ull << 32)) return fail(ctx, POLEE_ERR_UNSUPPORTED, "device layout build: too many tiles");

    // ---- tiles and dictionaries
    const uint32_t nwaves = std::min<uint32_t>(nseg, (uint32_t)std::max(1, 2 * ctx->num_cus));
    const uint32_t mixed_rows = (uint32_t)(W.bounds[6] - W.bounds[4]);
    DevBuf<uint32_t> stamps, t_s0, t_cols, t_dstart, dict_s, srec_ri, srec_n;
    DevBuf<SegOut> outs;
    POLEE_TRY(stamps.alloc(ctx, (size_t)nwaves * A.n));
    POLEE_TRY(t_s0.alloc(ctx, (size_t)h_tot[0] + 1));
    POLEE_TRY(t_cols.alloc(ctx, (size_t)h_tot[0] + 1));
    POLEE_TRY(t_dstart.alloc(ctx, (size_t)h_tot[0] + 1));
    POLEE_TRY(dict_s.alloc(ctx, (size_t)h_tot[1] + 1));
    POLEE_TRY(srec_ri.alloc(ctx, (size_t)mixed_rows + 1));
    POLEE_TRY(srec_n.alloc(ctx, (size_t)mixed_rows + 1));
    POLEE_TRY(outs.alloc(ctx, nseg));
    PD_HIP(hipMemsetAsync(stamps.p, 0, (size_t)nwaves * A.n * 4, stream));
    PD_HIP(hipMemsetAsync(nseg_d.p + 1, 0, 4, stream));
    S3Seq Q;
    Q.A = A; Q.segs = segs.p; Q.aux = aux.p; Q.nseg = nseg; Q.rlen = rlen.p; Q.st_start = st_start.p; Q.escan = escan.p;
    Q.stamps = stamps.p; Q.t_s0 = t_s0.p; Q.t_cols = t_cols.p; Q.t_dstart = t_dstart.p; Q.dict_s = dict_s.p;
    Q.srec_ri = srec_ri.p; Q.srec_n = srec_n.p; Q.outs = outs.p; Q.next_seg = nseg_d.p + 1;
    {
        static const int a1cap = getenv("POLEE_TILE_A1") ? atoi(getenv("POLEE_TILE_A1")) : PSELL_TILE_SLICES_A1;
        static const int a2cap = getenv("POLEE_TILE_A2") ? atoi(getenv("POLEE_TILE_A2")) : PSELL_TILE_SLICES_A2;
        static const int a2mcap = getenv("POLEE_TILE_A2M") ? std::min(atoi(getenv("POLEE_TILE_A2M")), 126) : PSELL_TILE_SLICES_A2M;
        Q.caps[PSELL_A1] = Q.caps[PSELL_A1M] = (uint32_t)std::min(a1cap, 252);
        Q.caps[PSELL_A2] = (uint32_t)std::min(a2cap, 126);
        Q.caps[PSELL_A2M] = (uint32_t)a2mcap;
        Q.caps[PSELL_BN] = (uint32_t)PSELL_TILE_SLICES_BN;
        Q.caps[PSELL_B] = (uint32_t)PSELL_TILE_SLICES_B;
    }
    hipLaunchKernelGGL(s3_tiles_kernel, dim3(nwaves), dim3(64), 0, stream, Q);
    POLEE_KERNEL_CHECK(ctx);
    std::vector<SegOut> h_outs(nseg);
    PD_HIP(hipMemcpyAsync(h_outs.data(), outs.p, sizeof(SegOut) * nseg, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (getenv("POLEE_BUILD_TIMING")) {  // where the tile walk's time goes: per segment
        std::vector<uint32_t> ord(nseg);
        for (uint32_t k = 0; k < nseg; ++k) ord[k] = k;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return h_outs[a].pad > h_outs[b].pad; });
        double sum = 0.0;
        for (uint32_t k = 0; k < nseg; ++k) sum += h_outs[k].pad * 1e-5;
        fprintf(stderr, "[psell device build]   . tile walk: %u segments on %u waves, %.2f ms of wave time in all; the longest:\n", nseg, nwaves, sum);
        for (uint32_t q = 0; q < std::min<uint32_t>(nseg, 6); ++q) {
            const uint32_t k = ord[q];
            fprintf(stderr, "[psell device build]       stream %u rows %u stretches %u tiles %u slices %u: %.2f ms\n", h_segs[k].stream, h_segs[k].rb - h_segs[k].ra,
                    h_aux[k].st1 - h_aux[k].st0, h_outs[k].ntiles, h_outs[k].nslices, h_outs[k].pad * 1e-5);
        }
    }

    // ---- the final numbering: segments concatenated in order (psell_build.cpp, "concatenate the fragments")
    std::vector<SegFinal> h_fin(nseg);
    uint64_t T = 0, S = 0, Dn = 0;
    {
        int last_stream = 0;
        auto stream_ends = [&](int st) {
            if (st == PSELL_A1) out.num_tiles_a1 = (int64_t)T;
            if (st == PSELL_A1M) out.num_tiles_a1m = (int64_t)T;
            if (st == PSELL_A2) out.num_tiles_a2 = (int64_t)T;
            if (st == PSELL_A2M) {
                out.num_tiles_a = (int64_t)T;
                out.num_slices_a = (int64_t)S;
            }
            if (st == PSELL_BN) out.num_tiles_s = (int64_t)T;
        };
        for (uint32_t k = 0; k < nseg; ++k) {
            for (; last_stream < (int)h_segs[k].stream; ++last_stream) stream_ends(last_stream);
            h_fin[k] = SegFinal{(uint32_t)T, (uint32_t)S, (uint32_t)Dn, 0u};
            T += h_outs[k].ntiles;
            S += h_outs[k].nslices;
            Dn += h_outs[k].ndict;
        }
        for (; last_stream < PSELL_B; ++last_stream) stream_ends(last_stream);
    }
    if (T >= (1ull << 32) || S >= (1ull << 32) || Dn >= (1ull << 32)) return fail(ctx, POLEE_ERR_UNSUPPORTED, "device layout build: index overflow");
    out.num_tiles = (int64_t)T;
    out.num_slices = (int64_t)S;
    DevBuf<SegFinal> fin;
    POLEE_TRY(fin.upload(ctx, h_fin.data(), nseg));
    DevBuf<uint32_t> tile_slice, tile_dict, tile_cols, tile_seg, dict, sl_ri, sl_n, sl_units, sl_tile, sl_long, slice_off;
    DevBuf<uint8_t> slice_flags, slice_w;
    DevBuf<unsigned long long> stats;
    POLEE_TRY(tile_slice.alloc(ctx, T + 1));
    POLEE_TRY(tile_dict.alloc(ctx, T + 1));
    POLEE_TRY(tile_cols.alloc(ctx, T + 1));
    POLEE_TRY(tile_seg.alloc(ctx, T + 1));
    POLEE_TRY(dict.alloc(ctx, Dn + 1));
    POLEE_TRY(sl_ri.alloc(ctx, S + 1));
    POLEE_TRY(sl_n.alloc(ctx, S + 1));
    POLEE_TRY(sl_units.alloc(ctx, S + 1));
    POLEE_TRY(sl_tile.alloc(ctx, S + 1));
    POLEE_TRY(sl_long.alloc(ctx, S + 1));
    POLEE_TRY(slice_off.alloc(ctx, S + 1));
    POLEE_TRY(slice_flags.alloc(ctx, S + 1));
    POLEE_TRY(slice_w.alloc(ctx, S + 1));
    POLEE_TRY(stats.alloc(ctx, 32));
    PD_HIP(hipMemsetAsync(dict.p, 0, (Dn + 1) * 4, stream));
    PD_HIP(hipMemsetAsync(stats.p, 0, 32 * 8, stream));
    PD_HIP(hipMemsetAsync(sl_units.p + S, 0, 4, stream));
    {
        const uint32_t tails[2] = {(uint32_t)S, (uint32_t)Dn};
        PD_HIP(hipMemcpyAsync(tile_slice.p + T, &tails[0], 4, hipMemcpyHostToDevice, stream));
        PD_HIP(hipMemcpyAsync(tile_dict.p + T, &tails[1], 4, hipMemcpyHostToDevice, stream));
        PD_HIP(hipStreamSynchronize(stream));
    }
    hipLaunchKernelGGL(s3_tiles_final_kernel, dim3(nseg), dim3(256), 0, stream, segs.p, aux.p, outs.p, fin.p, nseg, t_s0.p, t_cols.p,
                       t_dstart.p, dict_s.p, tile_slice.p, tile_dict.p, tile_cols.p, tile_seg.p, dict.p);
    POLEE_KERNEL_CHECK(ctx);
    S3Size Z;
    Z.A = A; Z.segs = segs.p; Z.aux = aux.p; Z.fin = fin.p; Z.rlen = rlen.p; Z.endpos = endpos.p; Z.srec_ri = srec_ri.p; Z.srec_n = srec_n.p;
    Z.tile_slice = tile_slice.p; Z.tile_seg = tile_seg.p; Z.num_tiles = (uint32_t)T; Z.has_ks = X.ks != nullptr;
    Z.sl_ri = sl_ri.p; Z.sl_n = sl_n.p; Z.sl_units = sl_units.p; Z.sl_tile = sl_tile.p; Z.sl_long = sl_long.p;
    Z.slice_flags = slice_flags.p; Z.slice_w = slice_w.p; Z.stats = stats.p;
    if (T) {
        hipLaunchKernelGGL(s3_size_kernel, dim3((uint32_t)T), dim3(64), 0, stream, Z);
        POLEE_KERNEL_CHECK(ctx);
    }
    // slice offsets in 128-byte units (64-bit sum first: the stream is limited to 2^29 units)
    DevBuf<uint64_t> off64;
    POLEE_TRY(off64.alloc(ctx, S + 1));
    PD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(sl_units.p, ToU64()), off64.p, (uint64_t)0, (size_t)S + 1, stream));
    uint64_t total_units = 0;
    PD_HIP(hipMemcpyAsync(&total_units, off64.p + S, 8, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (total_units >= (1ull << 29)) return fail(ctx, POLEE_ERR_UNSUPPORTED, "likelihood matrix: matrix too large (the slice stream is limited to 64 GiB)");
    PD_HIP(exclusive_sum(tmp, sl_units.p, slice_off.p, 0u, (size_t)S + 1, stream));
    D.data_bytes = (size_t)total_units * 128;
    POLEE_TRY(D.data.alloc(ctx, D.data_bytes + 2048));  // slack: the LDS-DMA stream reads whole 1 KiB pieces
    PD_HIP(hipMemsetAsync(D.data.p, 0, D.data_bytes + 2048, stream));
    if (X.ks) POLEE_TRY(D.slice_ks.alloc(ctx, (size_t)S * 64 + 1));
    DevBuf<uint32_t> row_order;
    if (want_debug) POLEE_TRY(row_order.alloc(ctx, (size_t)S * 64 + 1));
    S3Emit E;
    E.A = A; E.segs = segs.p; E.tile_seg = tile_seg.p; E.tile_dict = tile_dict.p; E.tile_cols = tile_cols.p; E.dict = dict.p;
    E.sl_ri = sl_ri.p; E.sl_n = sl_n.p; E.sl_tile = sl_tile.p; E.sl_long = sl_long.p; E.slice_off = slice_off.p;
    E.num_slices = (uint32_t)S; E.data = D.data.p; E.row_order = want_debug ? row_order.p : nullptr; E.slice_ks = X.ks ? D.slice_ks.p : nullptr;
    if (S) {
        hipLaunchKernelGGL(s3_emit_kernel, dim3((uint32_t)S), dim3(64), 0, stream, E);
        POLEE_KERNEL_CHECK(ctx);
        hipLaunchKernelGGL(s3_flags_kernel, dim3(((uint32_t)S + TB - 1) / TB), dim3(TB), 0, stream, (uint32_t)S, slice_flags.p, slice_off.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    // ---- the metadata back to the host (schedule, cost model, slot lists: loglik_finish_create)
    out.slice_off.resize(S + 1);
    out.tile_slice.resize(T + 1);
    out.tile_dict.resize(T + 1);
    out.tile_cols.resize(T);
    out.dict.resize(Dn);
    out.slice_flags.resize(S);
    out.slice_w.resize(S);
    unsigned long long h_stats[32];
    PD_HIP(hipMemcpyAsync(out.slice_off.data(), slice_off.p, (S + 1) * 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(out.tile_slice.data(), tile_slice.p, (T + 1) * 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(out.tile_dict.data(), tile_dict.p, (T + 1) * 4, hipMemcpyDeviceToHost, stream));
    if (T) PD_HIP(hipMemcpyAsync(out.tile_cols.data(), tile_cols.p, T * 4, hipMemcpyDeviceToHost, stream));
    if (Dn) PD_HIP(hipMemcpyAsync(out.dict.data(), dict.p, Dn * 4, hipMemcpyDeviceToHost, stream));
    if (S) PD_HIP(hipMemcpyAsync(out.slice_flags.data(), slice_flags.p, S, hipMemcpyDeviceToHost, stream));
    if (S) PD_HIP(hipMemcpyAsync(out.slice_w.data(), slice_w.p, S, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipMemcpyAsync(h_stats, stats.p, sizeof h_stats, hipMemcpyDeviceToHost, stream));
    if (want_debug) {
        out.data.resize(D.data_bytes);
        out.row_order.resize(S * 64);
        if (D.data_bytes) PD_HIP(hipMemcpyAsync(out.data.data(), D.data.p, D.data_bytes, hipMemcpyDeviceToHost, stream));
        if (S) PD_HIP(hipMemcpyAsync(out.row_order.data(), row_order.p, S * 64 * 4, hipMemcpyDeviceToHost, stream));
        if (X.ks) {
            out.slice_ks.resize(S * 64);
            if (S) PD_HIP(hipMemcpyAsync(out.slice_ks.data(), D.slice_ks.p, S * 64 * 4, hipMemcpyDeviceToHost, stream));
        }
    }
    PD_HIP(hipStreamSynchronize(stream));
    for (int q = 0; q < 6; ++q) {
        out.stream_rows[q] += (int64_t)h_stats[q];
        out.stream_nnz[q] += (int64_t)h_stats[8 + q];
        out.stream_bytes[q] += (int64_t)h_stats[16 + q];
    }
    out.padded_nnz += (int64_t)h_stats[24];
    for (uint64_t t = 0; t < T; ++t) {
        out.max_tile_cols = std::max<int32_t>(out.max_tile_cols, (int32_t)(out.tile_dict[t + 1] - out.tile_dict[t]));
        if (out.tile_cols[t] > (uint32_t)PSELL_TILE_COLS_TARGET) out.big_tiles.push_back((uint32_t)t);
    }
    for (int64_t s = 0; s < out.num_slices_a; ++s)
        if (!(out.slice_flags[s] & 1)) return fail(ctx, POLEE_ERR_HIP, "internal error: non-uniform slice in the uniform stream");
    return POLEE_OK;
}

// ================================================================= INPUT ====================================================
namespace {

__global__ void s0_minus1_u64_kernel(uint64_t N, const uint64_t *in, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = in[i] - 1u;
}
__global__ void s0_minus1_u32_kernel(uint64_t N, const uint32_t *in, uint32_t *out, uint32_t *err)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (in[i] < 1) atomicMax(err, 1u);
    out[i] = in[i] - 1u;
}
// X by columns (CSC, 1-based) -> by rows: entries counted per row, a STABLE sort by row keeps a row's transcripts ascending
__global__ void s0_rowkeys_kernel(uint64_t nnz, uint64_t m, const uint32_t *rowval, uint32_t *key, uint32_t *idx, uint32_t *counts, uint32_t *err)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const uint32_t r = rowval[k];
    if (r < 1 || (uint64_t)r > m) {
        atomicMax(err, 1u);
        key[k] = 0;
        idx[k] = (uint32_t)k;
        return;
    }
    key[k] = r - 1u;
    idx[k] = (uint32_t)k;
    atomicAdd(&counts[r - 1u], 1u);
}
// (two kernels: the columns need colptr and the sorted positions only, so they are found while the VALUES are still on their way up)
__global__ void s0_gather_col_kernel(uint64_t nnz, int64_t n, const uint64_t *colptr, const uint32_t *idx, uint32_t *col)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const uint32_t k = idx[i];
    // the transcript j with colptr[j] - 1 <= k < colptr[j + 1] - 1
    int64_t lo = 0, hi = n;  // first j with colptr[j] - 1 > k, minus one
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (colptr[mid] - 1u <= (uint64_t)k) lo = mid + 1; else hi = mid;
    }
    col[i] = (uint32_t)(lo - 1);
}
__global__ void s0_gather_val_kernel(uint64_t nnz, const uint32_t *idx, const float *nzval, float *val)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnz) val[i] = nzval[idx[i]];
}

}  // namespace

bool psell_device_enabled()
{
    static const bool on = [] {
        if (const char *e = getenv("POLEE_DEVICE_BUILD")) return atoi(e) != 0;
        // the host builder's experiment knobs are the host builder's
        for (const char *k : {"POLEE_PSELL_BINSH", "POLEE_PSELL_NO_SINGLES", "POLEE_PSELL_NO_RUNS", "POLEE_PSELL_MIN_UNIFORM", "POLEE_PSELL_NO_UNION",
                              "POLEE_PSELL_NO_MASK", "POLEE_PSELL_MASK_GAIN", "POLEE_PSELL_OVER_BUDGET", "POLEE_PSELL_RELAX", "POLEE_PSELL_RELAX0",
                              "POLEE_PSELL_MAX_GROUP", "POLEE_PSELL_NO_BN", "POLEE_PSELL_NO_CSR",
                              "POLEE_PSELL_CSR_MIN_SHARE", "POLEE_TILE_PER_WG"})
            if (getenv(k)) return false;
        return true;
    }();
    return on;
}

polee_status psell_device_rows_from_xt(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *tcolptr, const uint32_t *trowval, const float *tnzval,
                                       const int64_t *ks, bool on_device, PsellDevCSR &C)
{
    hipStream_t stream = ctx->stream;
    C.m = m;
    C.n = n;
    uint64_t last = 0;
    if (on_device) {
        PD_HIP(hipMemcpyAsync(&last, tcolptr + m, 8, hipMemcpyDeviceToHost, stream));
        PD_HIP(hipStreamSynchronize(stream));
    } else {
        last = tcolptr[m];
    }
    const uint64_t nnz = last - 1;
    DevBuf<uint64_t> up_ptr;
    DevBuf<uint32_t> up_col, err;
    const uint64_t *d_ptr = tcolptr;
    const uint32_t *d_col = trowval;
    if (!on_device) {
        POLEE_TRY(up_ptr.upload(ctx, tcolptr, (size_t)m + 1));
        POLEE_TRY(up_col.upload(ctx, trowval, (size_t)nnz));
        POLEE_TRY(C.val.upload(ctx, tnzval, (size_t)nnz));
        C.val_ptr = C.val.p;
        d_ptr = up_ptr.p;
        d_col = up_col.p;
    } else {
        C.val_ptr = tnzval;
    }
    if (ks) {
        if (on_device) C.ks_ptr = ks;
        else {
            POLEE_TRY(C.ks.upload(ctx, ks, (size_t)m));
            C.ks_ptr = C.ks.p;
        }
    }
    POLEE_TRY(C.rowptr.alloc(ctx, (size_t)m + 1));
    POLEE_TRY(C.col.alloc(ctx, (size_t)nnz + 1));
    POLEE_TRY(err.alloc(ctx, 1));
    PD_HIP(hipMemsetAsync(err.p, 0, 4, stream));
    hipLaunchKernelGGL(s0_minus1_u64_kernel, dim3((unsigned)((m + 1 + 255) / 256)), dim3(256), 0, stream, (uint64_t)m + 1, d_ptr, C.rowptr.p);
    POLEE_KERNEL_CHECK(ctx);
    if (nnz) {
        hipLaunchKernelGGL(s0_minus1_u32_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, nnz, d_col, C.col.p, err.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    uint32_t h_err = 0;
    PD_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (h_err) return fail(ctx, POLEE_ERR_BAD_ARG, "trowval must be 1-based");
    return POLEE_OK;
}

// X by columns ALREADY on the device (1-based colptr as 64-bit words, 1-based rowval: polee_devx, or this file's own upload) -> rows.
// own_rowval: this call's private copy, released as soon as the keys are made (peak memory); a shared copy stays.
polee_status psell_device_rows_from_dev_csc(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *d_cp, uint64_t nnz, const uint32_t *d_rowval,
                                            const float *d_nzval, const int64_t *ks, PsellDevCSR &C, DevBuf<uint32_t> *own_rowval,
                                            const float *late_nzval, DevBuf<float> *late_buf)
{
    hipStream_t stream = ctx->stream;
    C.m = m;
    C.n = n;
    if (nnz > 0 && m < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: rowval out of range");
    if (ks) {
        POLEE_TRY(C.ks.upload(ctx, ks, (size_t)m));
        C.ks_ptr = C.ks.p;
    }
    POLEE_TRY(C.rowptr.alloc(ctx, (size_t)m + 1));
    POLEE_TRY(C.col.alloc(ctx, (size_t)nnz + 1));
    POLEE_TRY(C.val.alloc(ctx, (size_t)nnz + 1));
    C.val_ptr = C.val.p;
    if (nnz == 0) {
        PD_HIP(hipMemsetAsync(C.rowptr.p, 0, ((size_t)m + 1) * 8, stream));
        PD_HIP(hipStreamSynchronize(stream));
        return POLEE_OK;
    }
    Scratch tmp(ctx);
    DevBuf<uint32_t> key, key_s, idx, idx_s, counts, err;
    POLEE_TRY(key.alloc(ctx, (size_t)nnz));
    POLEE_TRY(idx.alloc(ctx, (size_t)nnz));
    POLEE_TRY(counts.alloc(ctx, (size_t)m + 1));
    POLEE_TRY(err.alloc(ctx, 1));
    PD_HIP(hipMemsetAsync(counts.p, 0, ((size_t)m + 1) * 4, stream));
    PD_HIP(hipMemsetAsync(err.p, 0, 4, stream));
    hipLaunchKernelGGL(s0_rowkeys_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, nnz, (uint64_t)m, d_rowval, key.p, idx.p, counts.p,
                       err.p);
    POLEE_KERNEL_CHECK(ctx);
    uint32_t h_err = 0;
    PD_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, stream));
    PD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(counts.p, ToU64()), C.rowptr.p, (uint64_t)0, (size_t)m + 1, stream));
    PD_HIP(hipStreamSynchronize(stream));
    if (h_err) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: rowval out of range");
    if (own_rowval) own_rowval->release();
    counts.release();
    POLEE_TRY(key_s.alloc(ctx, (size_t)nnz));
    POLEE_TRY(idx_s.alloc(ctx, (size_t)nnz));
    unsigned bits = 1;
    while (bits < 32 && ((uint64_t)1 << bits) < (uint64_t)m) ++bits;
    {
        size_t bytes = 0;
        PD_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key_s.p, idx.p, idx_s.p, (size_t)nnz, 0, bits, stream));
        PD_HIP(tmp.need(bytes));
        PD_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key_s.p, idx.p, idx_s.p, (size_t)nnz, 0, bits, stream));
    }
    hipLaunchKernelGGL(s0_gather_col_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, nnz, n, d_cp, idx_s.p, C.col.p);
    POLEE_KERNEL_CHECK(ctx);
    if (late_nzval) {
        // the caller's values go up NOW, on a stream of their own, beside the sort and the column search queued above (8 ms of kernels
        // under a 20 ms copy at C2); the block came out of the cache behind a host wait, so nothing else of any stream touches it
        hipStream_t up = nullptr;
        PD_HIP(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        polee_status st = late_buf->alloc(ctx, (size_t)nnz);
        hipError_t e = hipSuccess;
        if (st == POLEE_OK) {
            e = hipMemcpyAsync(late_buf->p, late_nzval, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice, up);
            if (e == hipSuccess) e = hipStreamSynchronize(up);
        }
        (void)hipStreamDestroy(up);
        if (st != POLEE_OK) return st;
        PD_HIP(e);
        d_nzval = late_buf->p;
    }
    hipLaunchKernelGGL(s0_gather_val_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, nnz, idx_s.p, d_nzval, C.val.p);
    POLEE_KERNEL_CHECK(ctx);
    PD_HIP(hipStreamSynchronize(stream));
    return POLEE_OK;
}

// the checks polee_loglik_create makes of a 1-based colptr, widened to 64 bits; nnz out
polee_status psell_check_colptr(polee_ctx *ctx, int64_t n, const void *colptr, int colptr_bytes, std::vector<uint64_t> &cp, uint64_t &nnz)
{
    if (colptr_bytes != 4 && colptr_bytes != 8) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: colptr_bytes must be 4 or 8");
    cp.resize((size_t)n + 1);
    for (int64_t j = 0; j <= n; ++j)
        cp[(size_t)j] = colptr_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(colptr)[j] : reinterpret_cast<const uint64_t *>(colptr)[j];
    if (cp[0] != 1) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: colptr[0] must be 1 (1-based)");
    for (int64_t j = 0; j < n; ++j)
        if (cp[(size_t)j + 1] < cp[(size_t)j]) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: colptr is not monotone");
    nnz = cp[(size_t)n] - 1;
    return POLEE_OK;
}

polee_status psell_device_rows_from_csc(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                        const float *nzval, const int64_t *ks, PsellDevCSR &C, bool &needs_host)
{
    needs_host = false;
    C.m = m;
    C.n = n;
    std::vector<uint64_t> cp;
    uint64_t nnz = 0;
    POLEE_TRY(psell_check_colptr(ctx, n, colptr, colptr_bytes, cp, nnz));
    if (nnz >= (1ull << 32) - 1 || m >= ((int64_t)1 << 32) - 1) {
        needs_host = true;
        return POLEE_OK;
    }
    if (nnz > 0 && m < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: rowval out of range");
    DevBuf<uint64_t> d_cp;
    DevBuf<uint32_t> d_rowval;
    DevBuf<float> d_nzval;
    if (nnz) {
        POLEE_TRY(d_cp.upload(ctx, cp.data(), cp.size()));
        POLEE_TRY(d_rowval.upload(ctx, rowval, (size_t)nnz));
    }
    static const bool serial = getenv("POLEE_SERIAL_UPLOAD") != nullptr;  // (A/B: the values before the kernels, on the same stream)
    if (serial && nnz) POLEE_TRY(d_nzval.upload(ctx, nzval, (size_t)nnz));
    return psell_device_rows_from_dev_csc(ctx, m, n, d_cp.p, nnz, d_rowval.p, d_nzval.p, ks, C, &d_rowval, serial ? nullptr : nzval, &d_nzval);
}

// all three stages on the device.  needs_host: the layout is the host builder's to make (see psell_device_stage2).
polee_status psell_device_build(polee_ctx *ctx, const PsellDevIn &X, PsellHost &out, PsellDevOut &D, bool want_debug, bool &needs_host)
{
    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[psell device build] %-22s %.3f s\n", what, now() - t0);
        t0 = now();
    };
    PsellDevRuns R;
    PsellDevRowsOwned W;
    POLEE_TRY(psell_device_stage1(ctx, X, out, R, want_debug));
    lap("keys / sort / runs");
    POLEE_TRY(psell_device_stage2(ctx, X, R, out, W, needs_host));
    lap("packing");
    if (needs_host) return POLEE_OK;
    R.a1_rows.release(); R.a1_ends.release(); R.a2_rows.release(); R.a2_ends.release(); R.rb.release();
    POLEE_TRY(psell_device_stage3(ctx, X, W.view(), out, D, want_debug));
    lap("slices and tiles");
    return POLEE_OK;
}

}  // namespace polee
