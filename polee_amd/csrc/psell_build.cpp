// Host-side construction of the PSELL device layout (see loglik_internal.hpp).
// Plays the role of `Xt = SparseMatrixCSC(transpose(X))` in the reference
// (src/likelihood-approximation.jl:407): a one-off re-layout of X for the hot loop.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <numeric>

#include "loglik_internal.hpp"

namespace polee {

static inline uint32_t mix32(uint32_t h, uint32_t v)
{
    h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2);
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}

// LSD radix sort of (key, value) pairs by 64-bit key, 16 bits per pass, skipping
// passes whose digit is constant.
static void radix_sort_pairs(std::vector<uint64_t> &keys, std::vector<uint32_t> &vals)
{
    const size_t N = keys.size();
    if (N < 2) return;
    std::vector<uint64_t> k2(N);
    std::vector<uint32_t> v2(N);
    std::vector<size_t> hist(65536);
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = pass * 16;
        std::fill(hist.begin(), hist.end(), 0);
        for (size_t i = 0; i < N; ++i) hist[(keys[i] >> sh) & 0xffff]++;
        if (hist[(keys[0] >> sh) & 0xffff] == N) continue;
        size_t sum = 0;
        for (size_t d = 0; d < 65536; ++d) {
            size_t c = hist[d];
            hist[d] = sum;
            sum += c;
        }
        for (size_t i = 0; i < N; ++i) {
            size_t p = hist[(keys[i] >> sh) & 0xffff]++;
            k2[p] = keys[i];
            v2[p] = vals[i];
        }
        keys.swap(k2);
        vals.swap(v2);
    }
}

std::string build_psell(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                        const int64_t *ks, PsellHost &out)
{
    if (m < 0 || n < 1) return "bad matrix dimensions";
    if (n > (int64_t)1 << 31) return "more than 2^31 transcripts is not supported";
    out = PsellHost();
    out.m = m;
    out.n = n;
    out.nnz = (int64_t)rowptr[m];

    int binsh = 8;
    if (const char *e = getenv("POLEE_PSELL_BINSH")) binsh = std::max(0, std::min(24, atoi(e)));

    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[psell build] %-28s %.3f s\n", what, now() - t_prev);
        t_prev = now();
    };
    // 1. sort keys
    std::vector<uint64_t> keys;
    std::vector<uint32_t> rows;
    keys.reserve(m);
    rows.reserve(m);
    for (int64_t i = 0; i < m; ++i) {
        const uint64_t b = rowptr[i], e = rowptr[i + 1];
        if (e < b) return "row offsets are not monotone";
        const uint64_t len = e - b;
        if (len == 0) {
            ++out.empty_rows;
            continue;
        }
        if (len > (uint64_t)PSELL_MAX_TILE_COLS) return "a fragment is compatible with more than 1024 transcripts";
        uint32_t h = 0x12345u, first = col[b];
        for (uint64_t k = b; k < e; ++k) {
            if (col[k] >= (uint64_t)n) return "transcript index out of range";
            h = mix32(h, col[k]);
            first = std::min(first, col[k]);
        }
        out.max_row = std::max<int32_t>(out.max_row, (int32_t)len);
        const uint64_t key = ((uint64_t)(first >> binsh) << 40) | ((uint64_t)std::min<uint64_t>(len, 255) << 32) | h;
        keys.push_back(key);
        rows.push_back((uint32_t)i);
    }
    lap("keys");
    radix_sort_pairs(keys, rows);
    lap("radix sort");
    keys.clear();
    keys.shrink_to_fit();

    // 1b. runs of rows with the same transcript set.  Rows are split into three streams, each a
    // contiguous tile range processed by a differently specialised kernel launch:
    //   A1: uniform slices (every valid row of a slice has the same transcript set), set size <= 18
    //   A2: uniform slices, set size 19..28
    //   B : everything else (short runs, small run remainders, very wide rows)
    // A run of r rows contributes floor(r/64) whole slices to A, and its remainder as one more
    // (zero-padded) uniform slice when the remainder is at least 32 rows.
    std::vector<uint32_t> run_end;  // for stream A rows: index (in `rows`) one past the row's slice
    {
        auto same_set = [&](uint32_t r1, uint32_t r2) {
            const uint64_t l1 = rowptr[r1 + 1] - rowptr[r1], l2 = rowptr[r2 + 1] - rowptr[r2];
            return l1 == l2 && std::equal(col + rowptr[r1], col + rowptr[r1] + l1, col + rowptr[r2]);
        };
        std::vector<uint32_t> ra1, ra2, rb, e1, e2;
        size_t i = 0;
        while (i < rows.size()) {
            size_t j = i + 1;
            while (j < rows.size() && same_set(rows[i], rows[j])) ++j;
            const size_t r = j - i;
            const uint64_t len = rowptr[rows[i] + 1] - rowptr[rows[i]];
            size_t take = (r / PSELL_LANES) * PSELL_LANES;
            if (r - take >= (size_t)PSELL_MIN_UNIFORM_ROWS) take = r;
            if (len > (uint64_t)PSELL_WIDE_MAX) take = 0;
            std::vector<uint32_t> &dst = len <= (uint64_t)PSELL_NARROW_MAX ? ra1 : ra2;
            std::vector<uint32_t> &de = len <= (uint64_t)PSELL_NARROW_MAX ? e1 : e2;
            for (size_t q = 0; q < take; ++q) {
                dst.push_back(rows[i + q]);
                // slice boundary inside the run: after every 64 rows, and at the end of the taken part
                de.push_back((q + 1) % PSELL_LANES == 0 || q + 1 == take ? 1u : 0u);
            }
            rb.insert(rb.end(), rows.begin() + i + take, rows.begin() + j);
            i = j;
        }
        out.rows_a1 = (int64_t)ra1.size();
        out.rows_a = (int64_t)(ra1.size() + ra2.size());
        rows.swap(ra1);
        rows.insert(rows.end(), ra2.begin(), ra2.end());
        rows.insert(rows.end(), rb.begin(), rb.end());
        run_end.swap(e1);
        run_end.insert(run_end.end(), e2.begin(), e2.end());
    }

    lap("runs / stream split");
    // 2. greedy slices and tiles
    std::vector<uint32_t> col_stamp(n, 0);   // tile id + 1 in which the column was last registered
    std::vector<uint16_t> col_local(n, 0);
    out.data.reserve((size_t)((double)out.nnz * 6.6) + 4096);
    out.slice_off.push_back(0);
    out.tile_slice.push_back(0);
    out.tile_dict.push_back(0);
    if (ks) out.slice_ks.reserve(rows.size() + 64);
    out.row_order.reserve(rows.size() + 64);

    uint32_t tile_id = 1;        // stamp of the current tile
    uint32_t tile_cols = 0;      // dictionary size of the current tile
    uint32_t tile_nslices = 0;
    std::vector<uint32_t> slice_rows;  // original row ids of the slice being formed
    slice_rows.reserve(PSELL_LANES);
    std::vector<uint32_t> prev_pattern;  // transcript ids of the previous slice if it was uniform
    bool prev_uniform = false;
    int cur_stream = 0;  // 0 = A1, 1 = A2, 2 = B

    auto close_slice = [&]() {
        if (slice_rows.empty()) return;
        uint32_t w = 0;
        for (uint32_t r : slice_rows) w = std::max<uint32_t>(w, (uint32_t)(rowptr[r + 1] - rowptr[r]));
        const size_t base = out.data.size();
        if (cur_stream < 2) {
            // uniform streams: the 64 fragments share one transcript set, so the column ids are stored once:
            //   uint16 lcol[128] (256-byte header, w used) ; float val[w][64] (rows rotated, see below)
            // with row multiplicities (factored likelihood) a last row float ks[64] follows, so that they reach the
            // kernel through the same stream as the values
            out.data.resize(base + 256 + (size_t)w * 256 + (ks ? 256 : 0), 0);
            if (ks) {
                float *kr = reinterpret_cast<float *>(out.data.data() + base + 256 + (size_t)w * 256);
                for (size_t lane = 0; lane < slice_rows.size(); ++lane) kr[lane] = (float)ks[slice_rows[lane]];
            }
            uint16_t *hdr = reinterpret_cast<uint16_t *>(out.data.data() + base);
            float *vals = reinterpret_cast<float *>(out.data.data() + base + 256);
            const uint64_t b0 = rowptr[slice_rows[0]];
            for (uint32_t t = 0; t < w; ++t) hdr[t] = col_local[col[b0 + t]];
            for (size_t lane = 0; lane < slice_rows.size(); ++lane) {
                const uint64_t b = rowptr[slice_rows[lane]];
                // element r of row t sits at position (r + 4 t) & 63: bank-conflict-free operand reads for the MFMA phase
                for (uint32_t t = 0; t < w; ++t) vals[(size_t)t * 64 + ((lane + 4 * t) & 63)] = val[b + t];
            }
        } else {
            // mixed stream: float val[w][64]; uint16 lcol[w][64], padded to a multiple of 256 bytes
            out.data.resize(base + (((size_t)w * 384 + 255) & ~(size_t)255), 0);
            float *vals = reinterpret_cast<float *>(out.data.data() + base);
            uint16_t *lcols = reinterpret_cast<uint16_t *>(out.data.data() + base + (size_t)w * 256);
            for (size_t lane = 0; lane < slice_rows.size(); ++lane) {
                const uint32_t r = slice_rows[lane];
                const uint64_t b = rowptr[r], len = rowptr[r + 1] - b;
                uint16_t last = 0;
                for (uint32_t t = 0; t < w; ++t) {
                    if (t < len) {
                        vals[(size_t)t * 64 + lane] = val[b + t];
                        last = col_local[col[b + t]];
                    }
                    lcols[(size_t)t * 64 + lane] = last;  // padding repeats the last valid local id
                }
            }
        }
        for (size_t lane = 0; lane < PSELL_LANES; ++lane) {
            const bool valid = lane < slice_rows.size();
            out.row_order.push_back(valid ? slice_rows[lane] : 0xffffffffu);
            if (ks) out.slice_ks.push_back(valid ? (float)ks[slice_rows[lane]] : 0.0f);
        }
        // flags: bit0 = all 64 lanes hold rows with one and the same transcript set ("uniform"),
        //        bit1 = uniform and the same set as the previous slice of this tile ("continues")
        uint8_t flags = 0;
        {
            const uint32_t r0 = slice_rows[0];
            const uint64_t b0 = rowptr[r0], len0 = rowptr[r0 + 1] - b0;
            bool uni = true;
            for (size_t lane = 1; uni && lane < slice_rows.size(); ++lane) {
                const uint32_t r = slice_rows[lane];
                uni = (rowptr[r + 1] - rowptr[r] == len0) &&
                      std::equal(col + b0, col + b0 + len0, col + rowptr[r]);
            }
            if (uni) {
                flags |= 1;
                if (prev_uniform && tile_nslices > 0 && prev_pattern.size() == len0 &&
                    std::equal(prev_pattern.begin(), prev_pattern.end(), col + b0))
                    flags |= 2;
                prev_pattern.assign(col + b0, col + b0 + len0);
            }
            prev_uniform = uni;
        }
        out.slice_flags.push_back(flags);
        {
            const int st = cur_stream;
            out.stream_rows[st] += (int64_t)slice_rows.size();
            for (uint32_t r : slice_rows) out.stream_nnz[st] += (int64_t)(rowptr[r + 1] - rowptr[r]);
            out.stream_bytes[st] += (int64_t)(out.data.size() - base);
        }
        out.padded_nnz += (int64_t)w * 64;
        out.slice_off.push_back((uint32_t)(out.data.size() / 128));
        ++out.num_slices;
        ++tile_nslices;
        slice_rows.clear();
    };
    auto close_tile = [&]() {
        if (tile_nslices == 0) return;
        out.tile_slice.push_back((uint32_t)out.num_slices);
        out.tile_dict.push_back((uint32_t)out.dict.size());
        out.max_tile_cols = std::max<int32_t>(out.max_tile_cols, (int32_t)tile_cols);
        if (tile_cols > (uint32_t)PSELL_TILE_COLS_TARGET) out.big_tiles.push_back((uint32_t)out.num_tiles);
        ++out.num_tiles;
        ++tile_id;
        tile_cols = 0;
        tile_nslices = 0;
    };

    for (size_t ri = 0; ri < rows.size(); ++ri) {
        if ((int64_t)ri == out.rows_a1) {  // stream A1 ends here: start A2 on a fresh tile
            close_slice();
            close_tile();
            out.num_tiles_a1 = out.num_tiles;
            prev_uniform = false;
            cur_stream = 1;
        }
        if ((int64_t)ri == out.rows_a) {  // stream A ends here: start stream B on a fresh tile
            close_slice();
            close_tile();
            out.num_tiles_a = out.num_tiles;
            out.num_slices_a = out.num_slices;
            prev_uniform = false;
            cur_stream = 2;
        }
        const uint32_t r = rows[ri];
        const uint64_t b = rowptr[r], e = rowptr[r + 1];
        for (;;) {
            uint32_t fresh = 0;
            for (uint64_t k = b; k < e; ++k) fresh += col_stamp[col[k]] != tile_id;
            if (tile_cols + fresh <= (uint32_t)PSELL_TILE_COLS_TARGET || (tile_cols == 0 && slice_rows.empty())) break;
            // does not fit into the current tile: finish it (possibly with a partial slice)
            close_slice();
            close_tile();
        }
        for (uint64_t k = b; k < e; ++k) {
            const uint32_t c = col[k];
            if (col_stamp[c] != tile_id) {
                col_stamp[c] = tile_id;
                col_local[c] = (uint16_t)tile_cols++;
                out.dict.push_back(c);
            }
        }
        slice_rows.push_back(r);
        const bool boundary = (int64_t)ri < out.rows_a ? run_end[ri] != 0 : slice_rows.size() == PSELL_LANES;
        if (boundary) {
            close_slice();
            // small tiles for the two small streams (more workgroups, shorter tails)
            static const int a1cap = getenv("POLEE_TILE_A1") ? atoi(getenv("POLEE_TILE_A1")) : PSELL_TILE_SLICES_A1;
            const uint32_t cap = (int64_t)ri < out.rows_a1 ? (uint32_t)std::min(a1cap, 252)  // <= 63 slices per wave
                                 : (int64_t)ri < out.rows_a ? PSELL_TILE_SLICES_A2 : PSELL_TILE_SLICES_B;
            if (tile_nslices >= cap) close_tile();
        }
    }
    close_slice();
    close_tile();
    if ((int64_t)rows.size() == out.rows_a1) out.num_tiles_a1 = out.num_tiles;
    if ((int64_t)rows.size() == out.rows_a) {
        out.num_tiles_a = out.num_tiles;
        out.num_slices_a = out.num_slices;
    }
    if (out.rows_a1 == out.rows_a) out.num_tiles_a1 = std::min(out.num_tiles_a1, out.num_tiles_a);
    lap("slices and tiles");
    // 3. stream B: transposed copy of every tile (see PsellHost::tdata)
    {
        const int64_t ntb = out.num_tiles - out.num_tiles_a;
        out.ttile_group.assign(1, 0);
        out.tgroup_off.assign(1, 0);
        std::vector<uint32_t> cnt, start, piece_col, piece_beg, piece_len, order;
        std::vector<uint16_t> erow;
        std::vector<float> eval;
        for (int64_t tb = 0; tb < ntb; ++tb) {
            const int64_t tile = out.num_tiles_a + tb;
            const uint32_t d0 = out.tile_dict[tile], L = out.tile_dict[tile + 1] - d0;
            const uint32_t s0 = out.tile_slice[tile], s1 = out.tile_slice[tile + 1];
            for (uint32_t l = 0; l < L; ++l) col_local[out.dict[d0 + l]] = (uint16_t)l;
            // counting sort of the tile's entries by tile-local transcript id
            cnt.assign(L + 1, 0);
            for (uint32_t s = s0; s < s1; ++s)
                for (int lane = 0; lane < PSELL_LANES; ++lane) {
                    const uint32_t r = out.row_order[(size_t)s * 64 + lane];
                    if (r == 0xffffffffu) continue;
                    for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) ++cnt[col_local[col[k]] + 1];
                }
            for (uint32_t l = 0; l < L; ++l) cnt[l + 1] += cnt[l];
            start.assign(cnt.begin(), cnt.end());
            erow.resize(cnt[L]);
            eval.resize(cnt[L]);
            for (uint32_t s = s0; s < s1; ++s)
                for (int lane = 0; lane < PSELL_LANES; ++lane) {
                    const uint32_t r = out.row_order[(size_t)s * 64 + lane];
                    if (r == 0xffffffffu) continue;
                    for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                        const uint32_t p = start[col_local[col[k]]]++;
                        erow[p] = (uint16_t)((s - s0) * 64 + lane);
                        eval[p] = val[k];
                    }
                }
            // cut into pieces of <= PSELL_VCOL_CAP entries, longest first
            piece_col.clear(); piece_beg.clear(); piece_len.clear();
            for (uint32_t l = 0; l < L; ++l)
                for (uint32_t b = cnt[l]; b < cnt[l + 1]; b += PSELL_VCOL_CAP) {
                    piece_col.push_back(l);
                    piece_beg.push_back(b);
                    piece_len.push_back(std::min<uint32_t>(PSELL_VCOL_CAP, cnt[l + 1] - b));
                }
            order.resize(piece_col.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
            std::stable_sort(order.begin(), order.end(),
                             [&](uint32_t a, uint32_t b) { return piece_len[a] > piece_len[b]; });
            for (size_t g0 = 0; g0 < order.size(); g0 += PSELL_LANES) {
                const size_t g1 = std::min(order.size(), g0 + PSELL_LANES);
                const uint32_t width = piece_len[order[g0]];
                const size_t base = out.tdata.size();
                out.tdata.resize(base + 128 + (size_t)width * 384, 0);
                uint16_t *vcol = reinterpret_cast<uint16_t *>(out.tdata.data() + base);
                float *vval = reinterpret_cast<float *>(out.tdata.data() + base + 128);
                uint16_t *vrow = reinterpret_cast<uint16_t *>(out.tdata.data() + base + 128 + (size_t)width * 256);
                for (size_t i = g0; i < g1; ++i) {
                    const uint32_t pc = order[i], lane = (uint32_t)(i - g0);
                    vcol[lane] = (uint16_t)piece_col[pc];
                    for (uint32_t e = 0; e < piece_len[pc]; ++e) {
                        vval[(size_t)e * 64 + lane] = eval[piece_beg[pc] + e];
                        vrow[(size_t)e * 64 + lane] = erow[piece_beg[pc] + e];
                    }
                }
                out.tgroup_off.push_back((uint32_t)(out.tdata.size() / 128));
                ++out.num_groups;
            }
            out.ttile_group.push_back((uint32_t)out.num_groups);
        }
        if (out.tdata.size() / 128 >= (1ull << 32)) return "matrix too large (transposed copy of the mixed stream)";
        out.stream_bytes[2] += (int64_t)out.tdata.size();
    }
    for (int64_t s = 0; s < out.num_slices_a; ++s)
        if (!(out.slice_flags[s] & 1)) return "internal error: non-uniform slice in the uniform stream";
    lap("transposed copy (B)");
    if (out.data.size() / 128 >= (1ull << 30)) return "matrix too large (the slice stream is limited to 128 GiB)";
    // the two flag bits of slice s ride in the top bits of slice_off[s] (one scalar/lane load per slice)
    for (int64_t s = 0; s < out.num_slices; ++s) out.slice_off[s] |= (uint32_t)(out.slice_flags[s] & 3u) << 30;
    return "";
}

}  // namespace polee
