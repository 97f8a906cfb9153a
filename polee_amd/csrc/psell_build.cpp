// Host-side construction of the PSELL device layout (see loglik_internal.hpp).
// Plays the role of `Xt = SparseMatrixCSC(transpose(X))` in the reference
// (src/likelihood-approximation.jl:407): a one-off re-layout of X for the hot loop.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <numeric>
#include <thread>

#include "loglik_internal.hpp"

namespace polee {

static inline uint32_t mix32(uint32_t h, uint32_t v)
{
    h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2);
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}

// LSD radix sort of (key, value) pairs by 64-bit key, 16 bits per pass, skipping passes whose digit is
// constant; stable; histogram and scatter run on several host threads (fixed contiguous chunk per thread).
static void radix_sort_pairs(BVec<uint64_t> &keys, BVec<uint32_t> &vals)
{
    const size_t N = keys.size();
    if (N < 2) return;
    BVec<uint64_t> k2(N);
    BVec<uint32_t> v2(N);
    const unsigned T = N < ((size_t)1 << 16) ? 1u : std::min(host_threads(), 32u);
    std::vector<BVec<size_t>> hist(T, BVec<size_t>(65536));
    auto run = [&](auto &&f) {
        if (T == 1) {
            f(0u);
            return;
        }
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(f, t);
        f(0u);
        for (auto &th : pool) th.join();
    };
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = pass * 16;
        run([&](unsigned t) {
            BVec<size_t> &h = hist[t];
            std::fill(h.begin(), h.end(), 0);
            for (size_t i = N * t / T, e = N * (t + 1) / T; i < e; ++i) h[(keys[i] >> sh) & 0xffff]++;
        });
        size_t first_digit_count = 0;
        const size_t d0 = (keys[0] >> sh) & 0xffff;
        for (unsigned t = 0; t < T; ++t) first_digit_count += hist[t][d0];
        if (first_digit_count == N) continue;  // constant digit
        size_t sum = 0;
        for (size_t d = 0; d < 65536; ++d)
            for (unsigned t = 0; t < T; ++t) {
                const size_t c = hist[t][d];
                hist[t][d] = sum;
                sum += c;
            }
        run([&](unsigned t) {
            BVec<size_t> &h = hist[t];
            for (size_t i = N * t / T, e = N * (t + 1) / T; i < e; ++i) {
                const size_t p = h[(keys[i] >> sh) & 0xffff]++;
                k2[p] = keys[i];
                v2[p] = vals[i];
            }
        });
        keys.swap(k2);
        vals.swap(v2);
    }
}

namespace {
struct BuildClock {  // POLEE_BUILD_TIMING=1: wall time of every phase on stderr
    bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    double t_prev = now(), t_sub = now();
    void lap(const char *what)
    {
        if (timing) fprintf(stderr, "[psell build] %-28s %.3f s\n", what, now() - t_prev);
        t_prev = now();
    }
    void sublap(const char *what)  // (finer marks inside a phase; the phase's own line still covers all of it)
    {
        if (timing) fprintf(stderr, "[psell build]   . %-24s %.3f s\n", what, now() - t_sub);
        t_sub = now();
    }
};
}  // namespace

int psell_bin_shift()
{
    int binsh = 8;
    if (const char *e = getenv("POLEE_PSELL_BINSH")) binsh = std::max(0, std::min(24, atoi(e)));
    return binsh;
}

// The builder runs in three stages with explicit interfaces (loglik_internal.hpp), so that the device builder
// (psell_device.hip) can be checked against it stage by stage: any stage of either builder continues from the other's output.
//
// STAGE 1: sort keys, rows with one transcript collapsed (stream S), runs of rows with the same transcript set.
std::string psell_stage1(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                         const int64_t *ks, PsellHost &out, PsellRuns &R)
{
    if (m < 0 || n < 1) return "bad matrix dimensions";
    if (n > (int64_t)1 << 31) return "more than 2^31 transcripts is not supported";
    out = PsellHost();
    out.m = m;
    out.n = n;
    out.nnz = (int64_t)rowptr[m];
    R = PsellRuns();
    const int binsh = psell_bin_shift();
    BuildClock clk;
    auto lap = [&](const char *w) { clk.lap(w); };
    auto sublap = [&](const char *w) { clk.sublap(w); };
    // 1. sort keys (rows in parallel; empty rows get the key ~0 and are dropped afterwards).  Fragments with ONE compatible
    // transcript are collapsed here (stream S, loglik_internal.hpp): counted per transcript, their log X_ij summed per chunk
    // (the chunks' sums are added in chunk order: the constant does not depend on the number of threads), key ~0 too.
    BVec<uint64_t> keys((size_t)m);
    BVec<uint32_t> rows((size_t)m);
    {
        static const bool no_singles = getenv("POLEE_PSELL_NO_SINGLES") != nullptr;  // (A/B, tests: such rows stay in the slices)
        std::atomic<int> err{0};
        std::atomic<int64_t> empties{0}, singles{0};
        std::atomic<int32_t> max_row{0};
        const size_t KCH = (size_t)1 << 18;
        BVec<int64_t> scnt;
        if (!no_singles) scnt.assign((size_t)n, 0);
        BVec<double> chunk_log((size_t)((m + (int64_t)KCH - 1) / (int64_t)KCH) + 1, 0.0);
        parallel_chunks((size_t)m, KCH, [&](size_t lo, size_t hi, unsigned) {
            int64_t em = 0, sg = 0;
            int32_t mr = 0;
            double lsum = 0.0;
            for (size_t i = lo; i < hi; ++i) {
                const uint64_t b = rowptr[i], e = rowptr[i + 1];
                rows[i] = (uint32_t)i;
                if (e < b) { err = 1; keys[i] = ~0ull; continue; }
                const uint64_t len = e - b;
                if (len == 0) { ++em; keys[i] = ~0ull; continue; }
                if (len > (uint64_t)PSELL_MAX_TILE_COLS) { err = 2; keys[i] = ~0ull; continue; }
                if (len == 1 && !no_singles && col[b] < (uint64_t)n && val[b] > 0.0f && (!ks || ks[i] >= 0)) {
                    // (a non-positive X_ij stays a stored row: its log is the reference's -Inf / NaN, not ours to hide)
                    const int64_t k = ks ? ks[i] : 1;
                    __atomic_fetch_add(&scnt[col[b]], k, __ATOMIC_RELAXED);
                    lsum += (double)k * log((double)val[b]);
                    ++sg;
                    mr = std::max(mr, 1);
                    keys[i] = ~0ull;
                    continue;
                }
                uint32_t h = 0x12345u, first = col[b];
                for (uint64_t k = b; k < e; ++k) {
                    if (col[k] >= (uint64_t)n) err = 3;
                    if (k > b && col[k] <= col[k - 1]) err = 4;  // (the union / mask packing below walks a row's ids in order)
                    h = mix32(h, col[k]);
                }
                mr = std::max(mr, (int32_t)len);
                keys[i] = ((uint64_t)(first >> binsh) << 40) | ((uint64_t)std::min<uint64_t>(len, 255) << 32) | h;
            }
            empties += em;
            singles += sg;
            chunk_log[lo / KCH] = lsum;
            int32_t cur = max_row.load();
            while (mr > cur && !max_row.compare_exchange_weak(cur, mr)) {}
        });
        if (err == 1) return "row offsets are not monotone";
        if (err == 2) return "a fragment is compatible with more than 1024 transcripts";
        if (err == 3) return "transcript index out of range";
        if (err == 4) return "the transcript ids of a fragment must be strictly ascending (sorted, no duplicates)";
        out.empty_rows = empties;
        out.max_row = max_row;
        if (singles > 0) {
            out.single_cnt.resize((size_t)n);
            for (size_t j = 0; j < (size_t)n; ++j) out.single_cnt[j] = (float)scnt[j];
            for (double v : chunk_log) out.single_logsum += v;
            out.single_rows.reserve((size_t)singles);
            for (size_t i = 0; i < (size_t)m; ++i)
                if (keys[i] == ~0ull && rowptr[i + 1] - rowptr[i] == 1) out.single_rows.push_back((uint32_t)i);
            out.stream_rows[PSELL_S] = singles;
            out.stream_nnz[PSELL_S] = singles;
            out.stream_bytes[PSELL_S] = 4 * n;
        }
        if (out.empty_rows > 0 || singles > 0) {  // drop the empty and the collapsed rows (stable)
            size_t w = 0;
            for (size_t i = 0; i < (size_t)m; ++i)
                if (keys[i] != ~0ull) {
                    keys[w] = keys[i];
                    rows[w] = rows[i];
                    ++w;
                }
            keys.resize(w);
            rows.resize(w);
        }
    }
    lap("keys");
    radix_sort_pairs(keys, rows);
    lap("radix sort");
    clk.t_sub = BuildClock::now();
    keys.clear();
    keys.shrink_to_fit();

    // 1b. runs of rows with the same transcript set, and the split into the four streams (loglik_internal.hpp):
    //   A1 / A2: a run of r identical rows contributes floor(r / 64) whole slices, and its remainder as one more
    //            (zero-padded) slice when the remainder is at least 32 rows;
    //   the LEFTOVER rows (short runs, small remainders) are packed greedily, in the order of their first transcript,
    //            into groups of up to 64 rows whose UNION has <= 16 (<= 32 for rows of 17..32) transcripts.  A group is
    //            stored in the cheapest form: a dense union slice (A1 / A2), a masked slice (A1M; narrow class only) --
    //            or, when neither is below what its rows cost in the mixed stream, not as a group at all;
    //   B:       rows of more than 32 transcripts and the leftover rows of rejected groups.
    {
        auto same_set = [&](uint32_t r1, uint32_t r2) {
            const uint64_t l1 = rowptr[r1 + 1] - rowptr[r1], l2 = rowptr[r2 + 1] - rowptr[r2];
            return l1 == l2 && std::equal(col + rowptr[r1], col + rowptr[r1] + l1, col + rowptr[r2]);
        };
        // head[i]: row i starts a new run (comparisons in parallel)
        BVec<uint8_t> head(rows.size() + 1, 1);
        parallel_chunks(rows.size(), (size_t)1 << 18, [&](size_t lo, size_t hi, unsigned) {
            for (size_t i = std::max<size_t>(lo, 1); i < hi; ++i) head[i] = same_set(rows[i - 1], rows[i]) ? 0 : 1;
        });
        sublap("run heads");
        BVec<uint32_t> &rb = R.rb;
        // (on several host threads: chunks of rows that start at a run's head, each with its own output lists, which are
        // then joined in chunk order -- the result does not depend on the number of threads)
        struct Part {
            BVec<uint32_t> ra1, ra2, rb, e1, e2;
        };
        const size_t CH = (size_t)1 << 20;
        const size_t nparts = std::max<size_t>(1, (rows.size() + CH - 1) / CH);
        BVec<size_t> pstart(nparts + 1, rows.size());
        for (size_t p = 0; p < nparts; ++p) {
            size_t a = p * CH;
            while (a < rows.size() && !head[a]) ++a;  // (head[0] = 1)
            pstart[p] = a;
        }
        std::vector<Part> parts(nparts);
        static const bool no_runs = getenv("POLEE_PSELL_NO_RUNS") != nullptr;  // (experiments: every row goes through the packing of leftover rows)
        static const int min_uniform = getenv("POLEE_PSELL_MIN_UNIFORM") ? atoi(getenv("POLEE_PSELL_MIN_UNIFORM")) : PSELL_MIN_UNIFORM_ROWS;
        parallel_chunks(nparts, 1, [&](size_t plo, size_t phi, unsigned) {
            for (size_t p = plo; p < phi; ++p) {
                Part &P = parts[p];
                size_t i = pstart[p];
                const size_t end = pstart[p + 1];
                while (i < end) {
                    size_t j = i + 1;
                    while (j < rows.size() && !head[j]) ++j;
                    const size_t r = j - i;
                    const uint64_t len = rowptr[rows[i] + 1] - rowptr[rows[i]];
                    size_t take = (r / PSELL_LANES) * PSELL_LANES;
                    if (r - take >= (size_t)min_uniform) take = r;
                    if (len > (uint64_t)PSELL_WIDE_MAX || no_runs) take = 0;
                    BVec<uint32_t> &dst = len <= (uint64_t)PSELL_NARROW_MAX ? P.ra1 : P.ra2;
                    BVec<uint32_t> &de = len <= (uint64_t)PSELL_NARROW_MAX ? P.e1 : P.e2;
                    for (size_t q = 0; q < take; ++q) {
                        dst.push_back(rows[i + q]);
                        // slice boundary inside the run: after every 64 rows, and at the end of the taken part
                        de.push_back((q + 1) % PSELL_LANES == 0 || q + 1 == take ? 1u : 0u);
                    }
                    P.rb.insert(P.rb.end(), rows.begin() + i + take, rows.begin() + j);
                    i = j;
                }
            }
        });
        {
            size_t n1 = 0, n2 = 0, nb = 0;
            for (const Part &P : parts) {
                n1 += P.ra1.size();
                n2 += P.ra2.size();
                nb += P.rb.size();
            }
            R.a1_rows.reserve(n1); R.a1_ends.reserve(n1); R.a2_rows.reserve(n2); R.a2_ends.reserve(n2); rb.reserve(nb);
            for (Part &P : parts) {
                R.a1_rows.insert(R.a1_rows.end(), P.ra1.begin(), P.ra1.end());
                R.a1_ends.insert(R.a1_ends.end(), P.e1.begin(), P.e1.end());
                R.a2_rows.insert(R.a2_rows.end(), P.ra2.begin(), P.ra2.end());
                R.a2_ends.insert(R.a2_ends.end(), P.e2.begin(), P.e2.end());
                rb.insert(rb.end(), P.rb.begin(), P.rb.end());
                Part().ra1.swap(P.ra1);
            }
        }
        sublap("exact runs -> streams");
    }
    return "";
}

// STAGE 2: packing of the leftover rows, the mixed streams, the CSR last resort; the ordered row list of every sliced stream.
std::string psell_stage2(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                         const int64_t *ks, PsellRuns &R, PsellHost &out, PsellRows &W)
{
    W = PsellRows();
    BuildClock clk;
    auto sublap = [&](const char *w) { clk.sublap(w); };
    BVec<uint32_t> &rows = W.rows, &run_end = W.run_end, &row_gid = W.row_gid;
    BVec<uint8_t> &row_form = W.row_form;
    std::vector<BVec<uint32_t>> patterns;  // transcript set (union) of every group of leftover rows
    {
        struct RowList {
            BVec<uint32_t> rows, ends;
            BVec<uint8_t> form;
            BVec<uint32_t> gid;  // union / masked rows: index of the group's transcript set in `patterns`
        };
        RowList S1, S1M, S2, S2M;  // A1, A1M, A2, A2M
        S1.rows.swap(R.a1_rows);
        S1.ends.swap(R.a1_ends);
        S2.rows.swap(R.a2_rows);
        S2.ends.swap(R.a2_ends);
        S1.form.assign(S1.rows.size(), 0);
        S2.form.assign(S2.rows.size(), 0);
        S1.gid.assign(S1.rows.size(), 0);
        S2.gid.assign(S2.rows.size(), 0);
        BVec<uint32_t> rb;
        rb.swap(R.rb);
        // Packing of the leftover rows.  Rows are visited in the order of their first transcript (then length, then set:
        // equal sets stay neighbours); a row joins the open group while the union stays within the pass's width and the
        // group has fewer than 64 rows, a misfit is deferred once.  Two passes: unions of <= 16 over the rows of <= 16
        // transcripts, then unions of <= 32 over everything the first pass left (fragments of genes with many isoforms,
        // whose unions pass 16 after two or three rows) and the rows of 17..32.  A closed group of R rows with cnt_r
        // non-zeros, union w, longest row M costs 256 (w + 1) bytes as a dense union slice and 256 (M + 1) (w <= 16) or
        // 256 (M + 2) (w > 16) as a masked slice; it becomes a uniform slice of the cheaper form (the dense one when
        // masking would save less than `mask_gain` of its bytes: the masked forms cost the kernel ~5 more vector
        // instructions per transcript of the union) when that is no more than what CSR spends on its rows,
        // 8 sum(cnt_r) + 4 R; otherwise its rows go on (to the second pass, then to the mixed stream: 6 B per entry +
        // padding).  So the layout's bytes per non-zero stay below CSR's for any sparsity pattern, and the (slow) mixed
        // kernel only sees fragments that are unrelated to their neighbours.
        static const bool no_union = getenv("POLEE_PSELL_NO_UNION") != nullptr;
        static const bool no_mask = getenv("POLEE_PSELL_NO_MASK") != nullptr;
        // (round 4, measured at C2 size on one box, default 0.15 -> 0.5: per-entry dropout 0.1 / 0.3 -4.7 % / -3.4 % kernel time,
        // every fragment its own subset -6 %, tiled real fixture -1.3 %, generator as built -0.4 %: the kernel is bound by the
        // instructions it issues, not by bytes, and a masked slice costs ~5 more vector instructions per transcript of the
        // union -- it only pays when it halves the slice)
        static const double mask_gain = getenv("POLEE_PSELL_MASK_GAIN") ? atof(getenv("POLEE_PSELL_MASK_GAIN")) : 0.5;
        static const double over_budget = getenv("POLEE_PSELL_OVER_BUDGET") ? atof(getenv("POLEE_PSELL_OVER_BUDGET")) : 0.02;
        static const double relax = getenv("POLEE_PSELL_RELAX") ? atof(getenv("POLEE_PSELL_RELAX")) : 2.0;
        // (first pass: a narrow slice up to twice CSR's cost still beats what its rows meet further down -- a wide masked
        // slice, 4 - 10 x the cycles -- within the same global allowance; tiled real fixture -3 % kernel time, others +-0.7 %)
        static const double relax0 = getenv("POLEE_PSELL_RELAX0") ? atof(getenv("POLEE_PSELL_RELAX0")) : 2.0;
        static const size_t max_group = getenv("POLEE_PSELL_MAX_GROUP") ? (size_t)atoll(getenv("POLEE_PSELL_MAX_GROUP")) : (size_t)1 << 14;
        // (Round 4 tried masked narrow slices riding in the A1 tiles of their genomic bin -- a per-slice kind flag, the kernel's dense
        // loop followed by its masked loop per wave: no net gain, profiles/r04_mixed_tiles_ab.txt -- and round 5 removed it: the
        // second loop cost the streaming kernel 15 KB of code, which every input paid for)
        const size_t ks_rows = ks ? 1 : 0;
        if (!no_union && !rb.empty()) {
            BVec<uint64_t> k2(rb.size());
            for (size_t q = 0; q < rb.size(); ++q) k2[q] = ((uint64_t)col[rowptr[rb[q]]] << 32) | q;  // (columns ascend within a row)
            BVec<uint32_t> idx(rb.size());
            for (size_t q = 0; q < rb.size(); ++q) idx[q] = (uint32_t)q;
            radix_sort_pairs(k2, idx);
            BVec<uint32_t> cand(rb.size());
            for (size_t q = 0; q < rb.size(); ++q) cand[q] = rb[idx[q]];
            BVec<uint32_t> keep_b, pool, next_pool;
            // pass 0: unions of <= 16; pass 1: unions of <= 32 over what is left.  What finds no company in either goes to
            // the mixed streams: rows of <= 16 transcripts to BN, inside the persistent launch.  Rows of 17..32 would need
            // the per-tile kernel's extra launch (30 us for a handful of rows), so the second pass keeps their groups even
            // above CSR's cost, as long as the whole excess stays within `over_budget` of the matrix's CSR bytes.
            for (int pass_w = 0; pass_w < 2; ++pass_w) {
                const size_t cap = pass_w == 1 ? (size_t)PSELL_WIDE_MAX : (size_t)PSELL_NARROW_MAX;
                pool.clear();
                if (pass_w == 0) {
                    for (uint32_t r : cand) {
                        const uint64_t len = rowptr[r + 1] - rowptr[r];
                        if (len <= (uint64_t)PSELL_NARROW_MAX) pool.push_back(r);
                        else if (len <= (uint64_t)PSELL_WIDE_MAX) next_pool.push_back(r);
                        else keep_b.push_back(r);
                    }
                } else {
                    // the previous pass's rejects (pass 1: and the rows of 17..32), merged back into first-transcript order
                    BVec<uint64_t> k3(next_pool.size());
                    for (size_t q = 0; q < next_pool.size(); ++q) k3[q] = ((uint64_t)col[rowptr[next_pool[q]]] << 32) | q;
                    BVec<uint32_t> i3(next_pool.size());
                    for (size_t q = 0; q < next_pool.size(); ++q) i3[q] = (uint32_t)q;
                    radix_sort_pairs(k3, i3);
                    pool.resize(next_pool.size());
                    for (size_t q = 0; q < next_pool.size(); ++q) pool[q] = next_pool[i3[q]];
                    next_pool.clear();
                }
                // (chunks of PSELL_PACK_CHUNK candidate rows are packed independently on several threads -- on the device: one
                // wave each, psell_device.hip -- and joined in order: a group never spans two chunks, and the result does not
                // depend on the number of threads)
                double pool_csr_bytes = 0.0;
                for (uint32_t r : pool) pool_csr_bytes += 8.0 * (double)(rowptr[r + 1] - rowptr[r]) + 4.0;
                const double matrix_csr_bytes = 8.0 * (double)rowptr[m] + 4.0 * (double)m;
                struct UPart {
                    RowList dense1, masked1, dense2, masked2;
                    BVec<uint32_t> left;
                    std::vector<BVec<uint32_t>> pats;  // the sets of this part's groups (RowList::gid is local to the part)
                };
                const size_t UCH = (size_t)PSELL_PACK_CHUNK;
                const size_t nup = std::max<size_t>(1, (pool.size() + UCH - 1) / UCH);
                std::vector<UPart> uparts(nup);
                parallel_chunks(nup, 1, [&](size_t ulo, size_t uhi, unsigned) {
                    BVec<uint32_t> deferred, group, uni, tmp;
                    for (size_t up = ulo; up < uhi; ++up) {
                        UPart &U = uparts[up];
                        const size_t p0 = up * UCH, p1 = std::min(pool.size(), p0 + UCH);
                        // (second pass only) what the part's groups may spend above CSR's cost of their rows, so that a
                        // few fragments without company do not cost a pass the mixed stream's extra launch
                        // (over_budget of the whole matrix's CSR bytes, shared out by the parts' candidate rows)
                        double wide_allowance = pass_w == 1 ? PSELL_PACK_WIDE_RESERVE : 0.0;
                        double allowance = 0.0;
                        if (pass_w == 1 || relax0 > 1.0) {
                            for (size_t q = p0; q < p1; ++q) allowance += 8.0 * (double)(rowptr[pool[q] + 1] - rowptr[pool[q]]) + 4.0;
                            allowance *= over_budget * matrix_csr_bytes / std::max(pool_csr_bytes, 1.0);
                        }
                        // A GROUP = consecutive candidates under one union; it is cut into slices of 64 rows which all
                        // carry the group's union -- runs of slices with the same set, whose gradient stays in registers.
                        auto close_group = [&]() {
                            if (group.empty()) return;
                            const bool narrow = uni.size() <= (size_t)PSELL_NARROW_MAX;
                            const uint32_t pid = (uint32_t)U.pats.size();
                            bool any = false;
                            for (size_t c0 = 0; c0 < group.size(); c0 += PSELL_LANES) {
                                const size_t c1 = std::min(group.size(), c0 + (size_t)PSELL_LANES);
                                size_t total = 0, longest = 0;
                                for (size_t q = c0; q < c1; ++q) {
                                    const size_t c = (size_t)(rowptr[group[q] + 1] - rowptr[group[q]]);
                                    total += c;
                                    longest = std::max(longest, c);
                                }
                                const double dense_bytes = 256.0 * (double)(uni.size() + 1 + ks_rows);
                                const double masked_bytes = no_mask ? 1e30 : 256.0 * (double)(longest + (narrow ? 1 : 2) + ks_rows);
                                const double budget = 8.0 * (double)total + 4.0 * (double)(c1 - c0);  // (CSR's cost of these rows)
                                const double cost = std::min(dense_bytes, masked_bytes);
                                bool worth = cost <= budget;
                                // (second pass: a slice up to `relax` times CSR's cost is still better than what is left
                                // for its rows -- mixed tiles of some twenty unrelated fragments each --, and rows too long
                                // for stream BN are kept at any cost; both within the allowance)
                                if (!worth && ((pass_w == 1 && (cost <= relax * budget || longest > (size_t)PSELL_MIXED_NARROW_MAX)) ||
                                               (pass_w == 0 && cost <= relax0 * budget))) {
                                    const double over = cost - budget;
                                    if (over <= allowance) {
                                        allowance -= over;
                                        worth = true;
                                    } else if (pass_w == 1 && longest > (size_t)PSELL_MIXED_NARROW_MAX && over <= wide_allowance) {
                                        // (the part's own small reserve for rows too long for stream BN: with parts of 4 096
                                        // rows the proportional allowance alone left a few hundred of them to stream B -- a
                                        // launch of its own, 38 us per pass, for 0.003 % of the non-zeros)
                                        wide_allowance -= over;
                                        worth = true;
                                    }
                                }
                                // (Round 5 tried accepting a slice at the cost of the form it is STORED in -- the dense form, which the
                                // kernel prefers unless masking halves the slice, may cost more than the minimum the slice is accepted
                                // at -- and a separate gain for wide unions: more masked slices and more rows left to the mixed
                                // streams, the pass 0.06 - 0.12 ms SLOWER on two inputs, profiles/r05_mask_gain_sweep.txt.  Not kept.)
                                if (worth) {
                                    const bool masked = masked_bytes < (1.0 - mask_gain) * dense_bytes;
                                    RowList &dst = narrow ? (masked ? U.masked1 : U.dense1) : (masked ? U.masked2 : U.dense2);
                                    for (size_t q = c0; q < c1; ++q) {
                                        dst.rows.push_back(group[q]);
                                        dst.ends.push_back(q + 1 == c1 ? 1u : 0u);
                                        dst.form.push_back(masked ? 2 : 1);
                                        dst.gid.push_back(pid);
                                    }
                                    any = true;
                                } else {
                                    U.left.insert(U.left.end(), group.begin() + c0, group.begin() + c1);
                                }
                            }
                            if (any) U.pats.push_back(uni);
                            group.clear();
                            uni.clear();
                        };
                        deferred.clear();
                        for (int pass = 0; pass < 2; ++pass) {
                            const uint32_t *src = pass == 0 ? pool.data() + p0 : deferred.data();
                            const size_t cnt = pass == 0 ? p1 - p0 : deferred.size();
                            BVec<uint32_t> next_deferred;
                            for (size_t qi = 0; qi < cnt; ++qi) {
                                const uint32_t r = src[qi];
                                const uint32_t *cb = col + rowptr[r], *ce = col + rowptr[r + 1];
                                tmp.resize(uni.size() + (size_t)(ce - cb));
                                tmp.resize((size_t)(std::set_union(uni.begin(), uni.end(), cb, ce, tmp.begin()) - tmp.begin()));
                                // (a group that already fills a slice takes more rows only while that costs the matrix
                                // cores nothing: the same number of groups of four transcripts)
                                if (tmp.size() <= cap && (group.size() < (size_t)PSELL_LANES || (tmp.size() + 3) / 4 == (uni.size() + 3) / 4) &&
                                    group.size() < max_group) {
                                    uni.swap(tmp);
                                    group.push_back(r);
                                    continue;
                                }
                                if (pass == 0 && group.size() < 48 && uni.size() + 1 < cap) {
                                    next_deferred.push_back(r);  // an outlier (a neighbouring gene's isoform): second pass
                                    continue;
                                }
                                close_group();
                                uni.assign(cb, ce);
                                group.push_back(r);
                            }
                            close_group();
                            if (pass == 0) deferred.swap(next_deferred);
                        }
                    }
                });
                uint32_t gid_base = 0;
                auto append = [&](RowList &D, const RowList &U) {
                    D.rows.insert(D.rows.end(), U.rows.begin(), U.rows.end());
                    D.ends.insert(D.ends.end(), U.ends.begin(), U.ends.end());
                    D.form.insert(D.form.end(), U.form.begin(), U.form.end());
                    for (uint32_t g : U.gid) D.gid.push_back(g + gid_base);
                };
                for (UPart &U : uparts) {
                    gid_base = (uint32_t)patterns.size();
                    for (auto &pt : U.pats) patterns.push_back(std::move(pt));
                    append(S1, U.dense1);
                    append(S1M, U.masked1);
                    append(S2, U.dense2);
                    append(S2M, U.masked2);
                    BVec<uint32_t> &left = pass_w == 0 ? next_pool : keep_b;
                    left.insert(left.end(), U.left.begin(), U.left.end());
                }
            }
            rb.swap(keep_b);
        }
        sublap("packing of leftover rows");
        // the mixed streams -- BN: rows of <= 16 transcripts, B: the others -- each in the order of the rows' first
        // transcripts (small tile dictionaries), and inside every block of 1024 rows -- a tile's worth -- by descending
        // length (little padding inside a slice)
        BVec<uint32_t> rbn;
        if (!rb.empty()) {
            BVec<uint64_t> k2(rb.size());
            for (size_t q = 0; q < rb.size(); ++q) k2[q] = ((uint64_t)col[rowptr[rb[q]]] << 32) | q;
            BVec<uint32_t> idx(rb.size());
            for (size_t q = 0; q < rb.size(); ++q) idx[q] = (uint32_t)q;
            radix_sort_pairs(k2, idx);
            BVec<uint32_t> wide;
            static const bool no_bn = getenv("POLEE_PSELL_NO_BN") != nullptr;  // (experiments: every mixed row to the per-tile kernel)
            for (size_t q = 0; q < rb.size(); ++q) {
                const uint32_t r = rb[idx[q]];
                (rowptr[r + 1] - rowptr[r] <= (uint64_t)PSELL_MIXED_NARROW_MAX && !no_bn ? rbn : wide).push_back(r);
            }
            rb.swap(wide);
            const size_t BL = (size_t)PSELL_LANES * PSELL_TILE_SLICES_B;
            for (BVec<uint32_t> *lst : {&rbn, &rb})
                parallel_chunks((lst->size() + BL - 1) / BL, 16, [&](size_t lo, size_t hi, unsigned) {
                    for (size_t blk = lo; blk < hi; ++blk)
                        std::stable_sort(lst->begin() + blk * BL, lst->begin() + std::min(lst->size(), (blk + 1) * BL), [&](uint32_t r1, uint32_t r2) {
                            return rowptr[r1 + 1] - rowptr[r1] > rowptr[r2 + 1] - rowptr[r2];
                        });
                });
        }
        // Last resort, stream C: mixed tiles pay for a tile dictionary and for padding to the slice's longest row, and
        // fragments without any structure (a random sparse matrix: every row in other transcripts than its neighbours)
        // fill a 128-entry dictionary with less than one slice -- 15 - 20 bytes per non-zero.  The tiles the mixed
        // streams WOULD form are simulated here (same order, same closing rules); rows of a tile that would cost more
        // than CSR's 8 B per non-zero + 4 B per row are kept as they are, in CSR, for loglik_csr_kernel (lane = row,
        // global gathers and atomics).  So no input makes the layout larger than CSR.
        // (Only when that concerns a real share of the matrix -- more than `csr_min_share` (a tenth) of its non-zeros: a handful of
        // fragments without company stay in stream BN, inside the persistent launch, rather than cost every pass a launch
        // of their own.)
        sublap("mixed streams");
        static const bool no_csr = getenv("POLEE_PSELL_NO_CSR") != nullptr;
        static const double csr_min_share = getenv("POLEE_PSELL_CSR_MIN_SHARE") ? atof(getenv("POLEE_PSELL_CSR_MIN_SHARE")) : 0.10;
        BVec<uint32_t> rcsr;
        const BVec<uint32_t> rbn_all(rbn), rb_all(rb);
        if (!no_csr) {
            BVec<uint32_t> stamp((size_t)n, 0);
            uint32_t tile_stamp = 0;
            for (BVec<uint32_t> *lst : {&rbn, &rb}) {
                const size_t cap_slices = lst == &rbn ? (size_t)PSELL_TILE_SLICES_BN : (size_t)PSELL_TILE_SLICES_B;
                BVec<uint32_t> kept;
                size_t i = 0;
                while (i < lst->size()) {
                    // one simulated tile: rows i .. j-1
                    ++tile_stamp;
                    size_t j = i, dict = 0, nnz_t = 0, bytes = 0, slice_w = 0, in_slice = 0;
                    while (j < lst->size() && j - i < cap_slices * PSELL_LANES) {
                        const uint32_t r = (*lst)[j];
                        size_t fresh = 0;
                        for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) fresh += stamp[col[k]] != tile_stamp;
                        if (dict + fresh > (size_t)PSELL_TILE_COLS_TARGET && j > i) break;
                        for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) stamp[col[k]] = tile_stamp;
                        dict += fresh;
                        const size_t len = (size_t)(rowptr[r + 1] - rowptr[r]);
                        nnz_t += len;
                        slice_w = std::max(slice_w, len);
                        if (++in_slice == (size_t)PSELL_LANES) {
                            bytes += (slice_w * 384 + 255) & ~(size_t)255;
                            slice_w = in_slice = 0;
                        }
                        ++j;
                    }
                    if (in_slice) bytes += (slice_w * 384 + 255) & ~(size_t)255;
                    bytes += 4 * dict + 8;
                    if (bytes > 8 * nnz_t + 4 * (j - i))
                        rcsr.insert(rcsr.end(), lst->begin() + i, lst->begin() + j);
                    else
                        kept.insert(kept.end(), lst->begin() + i, lst->begin() + j);
                    i = j;
                }
                lst->swap(kept);
            }
        }
        {
            uint64_t csr_nnz = 0;
            for (uint32_t r : rcsr) csr_nnz += rowptr[r + 1] - rowptr[r];
            if (csr_nnz >= (1ull << 32)) return "more than 2^32 non-zeros without any structure are not supported";
            if ((double)csr_nnz < csr_min_share * (double)rowptr[m]) {  // not worth a launch: back to the mixed streams
                rcsr.clear();
                rbn = rbn_all;
                rb = rb_all;
            }
        }
        if (!rcsr.empty()) {
            std::sort(rcsr.begin(), rcsr.end());  // (the fragments' input order: neighbouring rows, neighbouring transcripts)
            out.csr_rowptr.assign(1, 0);
            for (uint32_t r : rcsr) {
                out.csr_col.insert(out.csr_col.end(), col + rowptr[r], col + rowptr[r + 1]);
                out.csr_val.insert(out.csr_val.end(), val + rowptr[r], val + rowptr[r + 1]);
                out.csr_rowptr.push_back((uint32_t)out.csr_col.size());
                if (ks) out.csr_ks.push_back((float)ks[r]);
            }
            out.csr_rows.assign(rcsr.begin(), rcsr.end());
            out.stream_rows[PSELL_C] = (int64_t)rcsr.size();
            out.stream_nnz[PSELL_C] = (int64_t)out.csr_col.size();
            out.stream_bytes[PSELL_C] = (int64_t)(8 * out.csr_col.size() + 4 * (rcsr.size() + 1));
        }
        out.rows_a1 = (int64_t)S1.rows.size();
        out.rows_a1m = out.rows_a1 + (int64_t)S1M.rows.size();
        out.rows_a2 = out.rows_a1m + (int64_t)S2.rows.size();
        out.rows_a = out.rows_a2 + (int64_t)S2M.rows.size();
        out.rows_s = out.rows_a + (int64_t)rbn.size();
        rows.swap(S1.rows);
        rows.insert(rows.end(), S1M.rows.begin(), S1M.rows.end());
        rows.insert(rows.end(), S2.rows.begin(), S2.rows.end());
        rows.insert(rows.end(), S2M.rows.begin(), S2M.rows.end());
        rows.insert(rows.end(), rbn.begin(), rbn.end());
        rows.insert(rows.end(), rb.begin(), rb.end());
        run_end.swap(S1.ends);
        run_end.insert(run_end.end(), S1M.ends.begin(), S1M.ends.end());
        run_end.insert(run_end.end(), S2.ends.begin(), S2.ends.end());
        run_end.insert(run_end.end(), S2M.ends.begin(), S2M.ends.end());
        row_form.swap(S1.form);
        row_form.insert(row_form.end(), S1M.form.begin(), S1M.form.end());
        row_form.insert(row_form.end(), S2.form.begin(), S2.form.end());
        row_form.insert(row_form.end(), S2M.form.begin(), S2M.form.end());
        row_gid.swap(S1.gid);
        row_gid.insert(row_gid.end(), S1M.gid.begin(), S1M.gid.end());
        row_gid.insert(row_gid.end(), S2.gid.begin(), S2.gid.end());
        row_gid.insert(row_gid.end(), S2M.gid.begin(), S2M.gid.end());
    }

    sublap("CSR last resort + totals");
    // the groups' transcript sets, flattened
    W.pat_ptr.assign(1, 0);
    {
        size_t tot = 0;
        for (const auto &pt : patterns) tot += pt.size();
        W.pat_col.reserve(tot);
        W.pat_ptr.reserve(patterns.size() + 1);
        for (const auto &pt : patterns) {
            W.pat_col.insert(W.pat_col.end(), pt.begin(), pt.end());
            W.pat_ptr.push_back((uint32_t)W.pat_col.size());
        }
    }
    clk.lap("packing / stream split");
    return "";
}

// STAGE 3: slices and tiles.
std::string psell_stage3(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                         const int64_t *ks, const PsellRows &W, PsellHost &out)
{
    BuildClock clk;
    auto lap = [&](const char *w) { clk.lap(w); };
    const BVec<uint32_t> &rows = W.rows, &run_end = W.run_end, &row_gid = W.row_gid;
    const BVec<uint8_t> &row_form = W.row_form;
    const uint32_t *pat_ptr = W.pat_ptr.data(), *pat_col = W.pat_col.data();
    // 2. greedy slices and tiles.  The three streams are cut into SEGMENTS of about a million rows (at slice
    // boundaries; every segment starts a fresh tile) which are laid out independently, on several host threads, and
    // concatenated afterwards.  The cut points depend on the data only, not on the number of threads.
    struct Segment {
        size_t ra, rb;  // rows[ra, rb)
        int stream;     // PSELL_A1 .. PSELL_B
        PsellHost frag;
    };
    std::vector<Segment> segs;
    {
        static const size_t seg_env = getenv("POLEE_PSELL_SEG_ROWS") ? (size_t)atoll(getenv("POLEE_PSELL_SEG_ROWS")) : 0;  // (tests)
        const size_t SEG_ROWS = seg_env >= 64 ? seg_env : (size_t)1 << 18;  // (a few hundred segments at BASELINE's C2: enough for ~50 host threads)
        const size_t bounds[PSELL_NSTREAMS + 1] = {0, (size_t)out.rows_a1, (size_t)out.rows_a1m, (size_t)out.rows_a2, (size_t)out.rows_a, (size_t)out.rows_s, rows.size()};
        for (int st = 0; st < PSELL_NSTREAMS; ++st) {
            size_t a = bounds[st];
            const size_t step = (st == PSELL_B || st == PSELL_BN) ? std::min(SEG_ROWS, (size_t)PSELL_MIXED_SEG_ROWS) : SEG_ROWS;
            while (a < bounds[st + 1]) {
                size_t e = std::min(bounds[st + 1], a + step);
                if (e < bounds[st + 1]) {
                    if (st != PSELL_B && st != PSELL_BN) {
                        while (e < bounds[st + 1] && !run_end[e - 1]) ++e;  // uniform streams: end on a closed slice
                    } else {  // mixed stream: whole tiles of 16 slices
                        const size_t tile_rows = (size_t)PSELL_LANES * PSELL_TILE_SLICES_B;
                        e = std::min(bounds[st + 1], a + ((e - a + tile_rows - 1) / tile_rows) * tile_rows);
                    }
                }
                segs.push_back(Segment{a, e, st, PsellHost()});
                a = e;
            }
        }
    }
    static const int a1cap_env = getenv("POLEE_TILE_A1") ? atoi(getenv("POLEE_TILE_A1")) : PSELL_TILE_SLICES_A1;
    static const int a2cap = getenv("POLEE_TILE_A2") ? atoi(getenv("POLEE_TILE_A2")) : PSELL_TILE_SLICES_A2;
    static const int a2mcap = getenv("POLEE_TILE_A2M") ? std::min(atoi(getenv("POLEE_TILE_A2M")), 126) : PSELL_TILE_SLICES_A2M;
    // A small sample gives the persistent launch's ~1000 workgroups only a few tiles each, and the dynamic schedule can
    // balance no finer than a tile (measured on the tiled real fixture, 3 500 tiles: slowest workgroup 1.29 x the mean).
    // So the narrow streams' tiles shrink -- never below 16 slices, four per wave -- until there are about
    // `tiles_per_wg` x 1024 of them.
    static const int tiles_per_wg = getenv("POLEE_TILE_PER_WG") ? atoi(getenv("POLEE_TILE_PER_WG")) : 0;
    int a1cap = a1cap_env;
    if (tiles_per_wg > 0) {
        const size_t slices_est = (rows.size() + PSELL_LANES - 1) / PSELL_LANES;
        const size_t want = (size_t)tiles_per_wg * 1024;
        a1cap = (int)std::max<size_t>(16, std::min<size_t>((size_t)a1cap_env, (slices_est + want - 1) / want));
    }
    const unsigned nthreads = host_threads();
    std::vector<BVec<uint32_t>> stamps(nthreads);
    std::vector<BVec<uint16_t>> locals(nthreads);
    BVec<uint32_t> next_tile_id(nthreads, 1);
    auto emit_segment = [&](Segment &sg, unsigned th) {
    PsellHost &out = sg.frag;  // (shadows the result: a segment fills its own fragment)
    if (stamps[th].empty()) {
        stamps[th].assign((size_t)n, 0);
        locals[th].assign((size_t)n, 0);
    }
    BVec<uint32_t> &col_stamp = stamps[th];   // tile id in which the column was last registered
    BVec<uint16_t> &col_local = locals[th];
    const int cur_stream = sg.stream;
    out.data.reserve((size_t)((double)(rowptr[m] / std::max<size_t>(rows.size(), 1)) * 6.6 * (double)(sg.rb - sg.ra)) + 4096);
    out.slice_off.push_back(0);
    out.tile_slice.push_back(0);
    out.tile_dict.push_back(0);
    if (ks) out.slice_ks.reserve(sg.rb - sg.ra + 64);
    out.row_order.reserve(sg.rb - sg.ra + 64);

    uint32_t &tile_id = next_tile_id[th];  // stamp of the current tile (unique per thread)
    uint32_t tile_cols = 0;      // dictionary size of the current tile
    uint32_t tile_nslices = 0;
    BVec<uint32_t> slice_rows;  // original row ids of the slice being formed
    slice_rows.reserve(PSELL_LANES);
    BVec<uint32_t> prev_pattern;  // transcript ids of the previous slice if it was uniform
    BVec<uint32_t> pattern;  // transcript set of the slice being closed (uniform streams)
    bool prev_uniform = false;
    int slice_form = 0;  // the slice being formed: 0 rows of one set, 1 dense union, 2 masked
    uint32_t slice_gid = 0;  // ... forms 1, 2: its group
    const bool uniform_stream = cur_stream != PSELL_B && cur_stream != PSELL_BN;
    // (round 4) the slices of a tile are QUEUED and emitted when the tile closes: in stream A1 the dense slices first, the
    // masked ones (leftover fragments riding in the tile) behind them -- the kernel runs its dense loop, then its masked
    // loop, over every wave's share of the tile
    struct PendingSlice {
        BVec<uint32_t> rows;
        int form;
        uint32_t gid;
    };
    std::vector<PendingSlice> pending;
    auto emit_slice = [&]() {
        if (slice_rows.empty()) return;
        uint32_t w = 0, longest = 0;
        for (uint32_t r : slice_rows) longest = std::max<uint32_t>(longest, (uint32_t)(rowptr[r + 1] - rowptr[r]));
        w = longest;
        if (uniform_stream) {  // the slice's transcript set: its rows' common set, or their union
            const uint32_t r0 = slice_rows[0];
            if (slice_form != 0)
                pattern.assign(pat_col + pat_ptr[slice_gid], pat_col + pat_ptr[slice_gid + 1]);  // the union of the slice's GROUP (a superset of its rows' sets)
            else
                pattern.assign(col + rowptr[r0], col + rowptr[r0 + 1]);
            w = (uint32_t)pattern.size();
        }
        const size_t base = out.data.size();
        uint32_t stored_rows = w;  // rows of 64 values the slice stores
        const bool masked_slice = cur_stream == PSELL_A1M || cur_stream == PSELL_A2M || (cur_stream == PSELL_A1 && slice_form == 2);
        const int stat_stream = masked_slice && cur_stream == PSELL_A1 ? PSELL_A1M : cur_stream;  // (accounting: by slice kind)
        if (masked_slice) {
            // masked slice: header rows of uint32 hw[64] -- one for unions of <= 16, two for 17..32: low half = bits
            // 0..15 (16..31) of the mask of the fragment in lane r; high half, r < 16: tile-local id of transcript r
            // (16 + r) of the union, PSELL_NO_COL past it -- then float val[i][64] = the i-th non-zero of the fragment in
            // lane r, i < longest row (+ the ks row)
            const size_t hrows = cur_stream == PSELL_A2M ? 2 : 1;
            stored_rows = longest;
            out.data.resize(base + 256 * hrows + (size_t)longest * 256 + (ks ? 256 : 0), 0);
            uint32_t *hw = reinterpret_cast<uint32_t *>(out.data.data() + base);
            float *vals = reinterpret_cast<float *>(out.data.data() + base + 256 * hrows);
            for (uint32_t t = 0; t < 16u * hrows; ++t) hw[(t >> 4) * 64 + (t & 15)] = (uint32_t)(t < w ? col_local[pattern[t]] : PSELL_NO_COL) << 16;
            for (size_t lane = 0; lane < slice_rows.size(); ++lane) {
                const uint64_t b = rowptr[slice_rows[lane]], e = rowptr[slice_rows[lane] + 1];
                uint32_t t = 0, mk = 0;
                for (uint64_t k = b; k < e; ++k) {
                    while (pattern[t] != col[k]) ++t;
                    mk |= 1u << t;
                    vals[(size_t)(k - b) * 64 + lane] = val[k];
                }
                hw[lane] |= mk & 0xffffu;
                if (hrows == 2) hw[64 + lane] |= mk >> 16;
            }
            if (ks) {
                float *kr = reinterpret_cast<float *>(out.data.data() + base + 256 * hrows + (size_t)longest * 256);
                for (size_t lane = 0; lane < slice_rows.size(); ++lane) kr[lane] = (float)ks[slice_rows[lane]];
            }
        } else if (uniform_stream) {
            // dense uniform slice: the 64 fragments share one transcript set, so the column ids are stored once:
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
            for (uint32_t t = 0; t < w; ++t) hdr[t] = col_local[pattern[t]];
            for (size_t lane = 0; lane < slice_rows.size(); ++lane) {
                const uint64_t b = rowptr[slice_rows[lane]], e = rowptr[slice_rows[lane] + 1];
                // element r of row t sits at position psell_row_pos(stream, t, r): bank-conflict-free operand reads for the MFMA phases
                if (slice_form == 0) {
                    for (uint32_t t = 0; t < w; ++t) vals[(size_t)t * 64 + psell_row_pos(cur_stream, t, (uint32_t)lane)] = val[b + t];
                } else {  // the row's entries at the positions of their transcripts in the union, zeros elsewhere
                    uint32_t t = 0;
                    for (uint64_t k = b; k < e; ++k) {
                        while (pattern[t] != col[k]) ++t;
                        vals[(size_t)t * 64 + psell_row_pos(cur_stream, t, (uint32_t)lane)] = val[k];
                    }
                }
            }
        } else {
            // mixed streams: float val[w][64]; uint16 lcol[w][64], padded to a multiple of 256 bytes; in stream BN with
            // multiplicities a row float ks[64] follows
            const size_t body = ((size_t)w * 384 + 255) & ~(size_t)255;
            out.data.resize(base + body + (ks && cur_stream == PSELL_BN ? 256 : 0), 0);
            if (ks && cur_stream == PSELL_BN) {
                float *kr = reinterpret_cast<float *>(out.data.data() + base + body);
                for (size_t lane = 0; lane < slice_rows.size(); ++lane) kr[lane] = (float)ks[slice_rows[lane]];
            }
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
        // flags: bit0 = all 64 lanes are stored under one transcript set ("uniform"),
        //        bit1 = uniform and the same set as the previous slice of this tile ("continues")
        uint8_t flags = masked_slice ? 4 : 0;
        if (uniform_stream) {
            flags |= 1;
            if (prev_uniform && tile_nslices > 0 && prev_pattern == pattern) flags |= 2;
            prev_pattern = pattern;
            prev_uniform = true;
        } else {
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
                prev_pattern.assign(col + b0, col + b0 + len0);
            }
            prev_uniform = uni;
        }
        out.slice_flags.push_back(flags);
        out.slice_w.push_back((uint8_t)std::min<uint32_t>(w, 255));
        {
            const int st = stat_stream;
            out.stream_rows[st] += (int64_t)slice_rows.size();
            for (uint32_t r : slice_rows) out.stream_nnz[st] += (int64_t)(rowptr[r + 1] - rowptr[r]);
            out.stream_bytes[st] += (int64_t)(out.data.size() - base);
        }
        out.padded_nnz += (int64_t)stored_rows * 64;
        out.slice_off.push_back((uint32_t)(out.data.size() / 128));
        ++out.num_slices;
        ++tile_nslices;
        slice_rows.clear();
        slice_form = 0;
    };
    auto close_slice = [&]() {
        if (slice_rows.empty()) return;
        pending.push_back(PendingSlice{slice_rows, slice_form, slice_gid});
        slice_rows.clear();
        slice_form = 0;
    };
    auto close_tile = [&]() {
        if (pending.empty()) return;
        for (int pass = 0; pass < 2; ++pass)  // (stable: runs of slices with one set stay together)
            for (PendingSlice &ps : pending) {
                const bool masked_narrow = cur_stream == PSELL_A1 && ps.form == 2;
                if ((int)masked_narrow != pass) continue;
                slice_rows.swap(ps.rows);
                slice_form = ps.form;
                slice_gid = ps.gid;
                emit_slice();
            }
        pending.clear();
        out.tile_slice.push_back((uint32_t)out.num_slices);
        out.tile_cols.push_back(tile_cols);
        while (out.dict.size() % PSELL_DICT_ALIGN) out.dict.push_back(0u);  // (never referenced by a slice)
        // (kernels that size their windows by tile_dict differences see the padded count)
        out.max_tile_cols = std::max<int32_t>(out.max_tile_cols, (int32_t)(out.dict.size() - out.tile_dict.back()));
        out.tile_dict.push_back((uint32_t)out.dict.size());
        if (tile_cols > (uint32_t)PSELL_TILE_COLS_TARGET) out.big_tiles.push_back((uint32_t)out.num_tiles);
        ++out.num_tiles;
        ++tile_id;
        tile_cols = 0;
        tile_nslices = 0;
    };

    for (size_t ri = sg.ra; ri < sg.rb; ++ri) {
        const uint32_t r = rows[ri];
        // the transcripts this row needs in the tile's dictionary: its own, or -- a row of a union / masked group -- the
        // whole set of its group (the slice's header lists every one of them)
        const bool grouped = uniform_stream && row_form[ri] != 0;
        const uint32_t *cb = grouped ? pat_col + pat_ptr[row_gid[ri]] : col + rowptr[r];
        const uint32_t *ce = grouped ? pat_col + pat_ptr[row_gid[ri] + 1] : col + rowptr[r + 1];
        for (;;) {
            uint32_t fresh = 0;
            for (const uint32_t *c = cb; c < ce; ++c) fresh += col_stamp[*c] != tile_id;
            if (tile_cols + fresh <= (uint32_t)PSELL_TILE_COLS_TARGET || (tile_cols == 0 && slice_rows.empty() && pending.empty())) break;
            // does not fit into the current tile: finish it (possibly with a partial slice)
            close_slice();
            close_tile();
        }
        for (const uint32_t *cp = cb; cp < ce; ++cp) {
            const uint32_t c = *cp;
            if (col_stamp[c] != tile_id) {
                col_stamp[c] = tile_id;
                col_local[c] = (uint16_t)tile_cols++;
                out.dict.push_back(c);
            }
        }
        slice_rows.push_back(r);
        if (uniform_stream) {
            slice_form = std::max<int>(slice_form, row_form[ri]);
            slice_gid = row_gid[ri];
        }
        const bool boundary = uniform_stream ? run_end[ri] != 0 : slice_rows.size() == PSELL_LANES;
        if (boundary) {
            close_slice();
            // small tiles for the two small streams (more workgroups, shorter tails)
            const uint32_t cap = cur_stream == PSELL_A1 || cur_stream == PSELL_A1M ? (uint32_t)std::min(a1cap, 252)  // <= 63 slices per wave
                                 : cur_stream == PSELL_A2 ? (uint32_t)std::min(a2cap, 126) : cur_stream == PSELL_A2M ? (uint32_t)a2mcap
                                 : cur_stream == PSELL_BN ? (uint32_t)PSELL_TILE_SLICES_BN : (uint32_t)PSELL_TILE_SLICES_B;
            if (pending.size() >= (size_t)cap) close_tile();
        }
    }
    close_slice();
    close_tile();
    };  // emit_segment
    {
        std::atomic<size_t> next{0};
        auto worker = [&](unsigned th) {
            for (size_t i = next++; i < segs.size(); i = next++) emit_segment(segs[i], th);
        };
        std::vector<std::thread> pool;
        const unsigned nt = (unsigned)std::min<size_t>(nthreads, std::max<size_t>(segs.size(), 1));
        for (unsigned th = 1; th < nt; ++th) pool.emplace_back(worker, th);
        worker(0);
        for (auto &t : pool) t.join();
    }
    lap("slices and tiles (segments)");
    // concatenate the fragments
    {
        size_t tot_data = 0, tot_slices = 0, tot_tiles = 0, tot_dict = 0;
        for (const Segment &sg : segs) {
            tot_data += sg.frag.data.size();
            tot_slices += (size_t)sg.frag.num_slices;
            tot_tiles += (size_t)sg.frag.num_tiles;
            tot_dict += sg.frag.dict.size();
        }
        out.data.reserve(tot_data + 4096);  // (the caller appends slack for the kernel's whole-piece reads: no reallocation)
        out.data.resize(tot_data);
        out.slice_off.assign(1, 0);
        out.tile_slice.assign(1, 0);
        out.tile_dict.assign(1, 0);
        out.slice_off.reserve(tot_slices + 1);
        out.tile_slice.reserve(tot_tiles + 1);
        out.tile_dict.reserve(tot_tiles + 1);
        out.tile_cols.reserve(tot_tiles);
        out.dict.reserve(tot_dict);
        out.slice_flags.reserve(tot_slices);
        out.slice_w.reserve(tot_slices);
        out.row_order.reserve(tot_slices * 64);
        if (ks) out.slice_ks.reserve(tot_slices * 64);
        BVec<size_t> data_base(segs.size());
        size_t dbase = 0;
        int last_stream = 0;
        auto stream_ends = [&](int st) {  // stream `st` ends at the current tile / slice count
            if (st == PSELL_A1) out.num_tiles_a1 = out.num_tiles;
            if (st == PSELL_A1M) out.num_tiles_a1m = out.num_tiles;
            if (st == PSELL_A2) out.num_tiles_a2 = out.num_tiles;
            if (st == PSELL_A2M) {
                out.num_tiles_a = out.num_tiles;
                out.num_slices_a = out.num_slices;
            }
            if (st == PSELL_BN) out.num_tiles_s = out.num_tiles;
        };
        for (size_t si = 0; si < segs.size(); ++si) {
            Segment &sg = segs[si];
            PsellHost &f = sg.frag;
            // stream boundaries in tile / slice numbering
            for (; last_stream < sg.stream; ++last_stream) stream_ends(last_stream);
            data_base[si] = dbase;
            const uint32_t unit_base = (uint32_t)(dbase / 128), slice_base = (uint32_t)out.num_slices,
                           dict_base = (uint32_t)out.dict.size(), tile_base = (uint32_t)out.num_tiles;
            if (dbase / 128 + f.data.size() / 128 >= (1ull << 29)) return "matrix too large (the slice stream is limited to 64 GiB)";
            for (size_t q = 1; q < f.slice_off.size(); ++q) out.slice_off.push_back(f.slice_off[q] + unit_base);
            for (size_t q = 1; q < f.tile_slice.size(); ++q) out.tile_slice.push_back(f.tile_slice[q] + slice_base);
            for (size_t q = 1; q < f.tile_dict.size(); ++q) out.tile_dict.push_back(f.tile_dict[q] + dict_base);
            out.dict.insert(out.dict.end(), f.dict.begin(), f.dict.end());
            out.tile_cols.insert(out.tile_cols.end(), f.tile_cols.begin(), f.tile_cols.end());
            out.slice_flags.insert(out.slice_flags.end(), f.slice_flags.begin(), f.slice_flags.end());
            out.slice_w.insert(out.slice_w.end(), f.slice_w.begin(), f.slice_w.end());
            out.row_order.insert(out.row_order.end(), f.row_order.begin(), f.row_order.end());
            if (ks) out.slice_ks.insert(out.slice_ks.end(), f.slice_ks.begin(), f.slice_ks.end());
            for (uint32_t bt : f.big_tiles) out.big_tiles.push_back(bt + tile_base);
            for (int q = 0; q < PSELL_NSTREAMS; ++q) {
                out.stream_rows[q] += f.stream_rows[q];
                out.stream_nnz[q] += f.stream_nnz[q];
                out.stream_bytes[q] += f.stream_bytes[q];
            }
            out.padded_nnz += f.padded_nnz;
            out.max_tile_cols = std::max(out.max_tile_cols, f.max_tile_cols);
            out.num_slices += f.num_slices;
            out.num_tiles += f.num_tiles;
            dbase += f.data.size();
        }
        for (; last_stream < PSELL_B; ++last_stream) stream_ends(last_stream);
        // the bulk copy in parallel
        parallel_chunks(segs.size(), 1, [&](size_t lo, size_t hi, unsigned) {
            for (size_t si = lo; si < hi; ++si) {
                auto &d = segs[si].frag.data;
                if (!d.empty()) memcpy(out.data.data() + data_base[si], d.data(), d.size());
                decltype(segs[si].frag.data)().swap(d);
            }
        });
        segs.clear();
    }
    lap("slices and tiles");
    for (int64_t s = 0; s < out.num_slices_a; ++s)
        if (!(out.slice_flags[s] & 1)) return "internal error: non-uniform slice in the uniform stream";
    if (out.data.size() / 128 >= (1ull << 29)) return "matrix too large (the slice stream is limited to 64 GiB)";
    // the two flag bits of slice s ride in the top bits of slice_off[s] (one scalar/lane load per slice)
    for (int64_t s = 0; s < out.num_slices; ++s)
        out.slice_off[s] |= ((uint32_t)(out.slice_flags[s] & 3u) << 30) | ((uint32_t)((out.slice_flags[s] >> 2) & 1u) << PSELL_FLAG_MASKED_BIT);
    return "";
}

std::string build_psell(int64_t m, int64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                        const int64_t *ks, PsellHost &out)
{
    PsellRuns R;
    PsellRows W;
    std::string err = psell_stage1(m, n, rowptr, col, val, ks, out, R);
    if (err.empty()) err = psell_stage2(m, n, rowptr, col, val, ks, R, out, W);
    if (err.empty()) err = psell_stage3(m, n, rowptr, col, val, ks, W, out);
    return err;
}

}  // namespace polee
