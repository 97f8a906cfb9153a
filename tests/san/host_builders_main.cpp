// Sanitizer driver for the HOST builders of libpolee_hip: the layout builder (psell_build.cpp, through
// polee_debug_psell_build: csc_to_csr's partitioned transposition, keys / radix sort / runs, the two packing passes, slice
// emission in parallel chunks), the tree heuristic in the reference's order (polee_hclust) and in rounds on all host threads
// (polee_hclust_parallel: unions, candidate lists and list updates under striped spin locks).  Built host-only
// (--offload-host-only) with -fsanitize=address,undefined or -fsanitize=thread by `make -C polee_amd/csrc sanitize SAN=...`
// and run by tests/test_sanitizers.py; it touches no GPU.  Exit code 0 = every call succeeded and its output is well-formed
// (the sanitizer itself aborts on a finding).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/polee_hip.h"
#include "../../include/polee_hip_debug.h"

struct Csc {
    int64_t m, n;
    std::vector<uint64_t> colptr;  // 1-based
    std::vector<uint32_t> rowval;  // 1-based
    std::vector<float> nzval;
};

// kind 0: genes with compatibility patterns (runs); 1: every fragment its own subset; 2: unstructured; 3: wide genes + strays
// + empty fragments
static Csc make(int kind, int64_t n, int64_t m, uint64_t seed)
{
    std::mt19937_64 rng(seed);
    std::vector<std::vector<uint32_t>> rows((size_t)m);
    std::vector<int64_t> gene_start;
    for (int64_t j = 0; j < n;) {
        gene_start.push_back(j);
        j += 1 + (int64_t)(rng() % (kind == 3 ? 40 : 12));
    }
    gene_start.push_back(n);
    const size_t G = gene_start.size() - 1;
    for (int64_t i = 0; i < m; ++i) {
        auto &r = rows[(size_t)i];
        if (kind == 3 && rng() % 50 == 0) continue;  // an empty fragment
        if (kind == 2) {
            const int k = 1 + (int)(rng() % 6);
            for (int t = 0; t < k; ++t) r.push_back((uint32_t)(rng() % (uint64_t)n));
        } else {
            const size_t g = (size_t)(rng() % G);
            const int64_t a = gene_start[g], b = std::min<int64_t>(gene_start[g + 1], n);
            if (kind == 0) {
                const uint64_t pat = 1 + (g * 7919 + rng() % 5) * 2654435761ull;
                for (int64_t j = a; j < b; ++j)
                    if ((pat >> ((j - a) % 60)) & 1) r.push_back((uint32_t)j);
                if (r.empty()) r.push_back((uint32_t)a);
            } else {
                for (int64_t j = a; j < b; ++j)
                    if (rng() % 4) r.push_back((uint32_t)j);
                if (r.empty()) r.push_back((uint32_t)a);
                if (kind == 3 && rng() % 20 == 0) r.push_back((uint32_t)(rng() % (uint64_t)n));  // a stray
            }
        }
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());
    }
    Csc X;
    X.m = m;
    X.n = n;
    X.colptr.assign((size_t)n + 1, 0);
    for (auto &r : rows)
        for (uint32_t j : r) ++X.colptr[(size_t)j + 1];
    X.colptr[0] = 1;
    for (int64_t j = 0; j < n; ++j) X.colptr[(size_t)j + 1] += X.colptr[(size_t)j];
    const size_t nnz = (size_t)(X.colptr[(size_t)n] - 1);
    X.rowval.resize(nnz);
    X.nzval.resize(nnz);
    std::vector<uint64_t> cur(X.colptr.begin(), X.colptr.end() - 1);
    std::uniform_real_distribution<float> u(1e-6f, 1e-2f);
    for (int64_t i = 0; i < m; ++i)
        for (uint32_t j : rows[(size_t)i]) {
            const size_t p = (size_t)(cur[j]++ - 1);
            X.rowval[p] = (uint32_t)(i + 1);
            X.nzval[p] = u(rng);
        }
    return X;
}

static int check_tree(const char *what, int64_t n, const std::vector<int32_t> &par, const std::vector<int32_t> &js)
{
    // serialised tree (src/ptt.jl:89-116): 2n-1 nodes, node 1 the root (parent 0), parents precede children, n leaves
    int64_t leaves = 0;
    for (size_t i = 0; i < par.size(); ++i) {
        if (i == 0 ? par[i] != 0 : (par[i] < 1 || par[i] > (int32_t)i)) {
            fprintf(stderr, "%s: node %zu has parent %d\n", what, i + 1, par[i]);
            return 1;
        }
        if (js[i] != 0) ++leaves;
    }
    if (leaves != n) {
        fprintf(stderr, "%s: %lld leaves, expected %lld\n", what, (long long)leaves, (long long)n);
        return 1;
    }
    return 0;
}

int main(int argc, char **argv)
{
    const int64_t scale = argc > 1 ? atoll(argv[1]) : 1;
    setenv("POLEE_HOST_THREADS", "8", 0);
    int bad = 0;
    for (int kind = 0; kind < 4; ++kind) {
        const int64_t n = 3000 * scale, m = 200000 * scale;
        Csc X = make(kind, n, m, 1234 + (uint64_t)kind);
        std::vector<int64_t> ks((size_t)m);
        for (int64_t i = 0; i < m; ++i) ks[(size_t)i] = 1 + (i * 2654435761ll) % 5;
        for (int with_ks = 0; with_ks < 2; ++with_ks) {
            polee_psell_debug *h = nullptr;
            polee_status st = polee_debug_psell_build(X.m, X.n, X.colptr.data(), 8, X.rowval.data(), X.nzval.data(),
                                                      with_ks ? ks.data() : nullptr, &h);
            if (st != POLEE_OK) {
                fprintf(stderr, "layout, kind %d: status %d: %s\n", kind, (int)st, polee_last_error(nullptr));
                ++bad;
                continue;
            }
            polee_psell_view v;
            if (polee_debug_psell_view(h, &v) != POLEE_OK || v.m != X.m || v.n != X.n) {
                fprintf(stderr, "layout, kind %d: bad view\n", kind);
                ++bad;
            }
            // every stored fragment appears once: slices' lanes + CSR rows + collapsed single-transcript rows + empties
            std::vector<uint8_t> seen((size_t)m, 0);
            int64_t dup = 0, cnt = 0;
            for (int64_t s = 0; s < v.num_slices * 64; ++s)
                if (v.row_order[s] != ~0u) dup += seen[v.row_order[s]]++, ++cnt;
            for (int64_t r = 0; r < v.csr_num_rows; ++r) dup += seen[v.csr_rows[r]]++, ++cnt;
            for (int64_t r = 0; r < v.single_num_rows; ++r) dup += seen[v.single_rows[r]]++, ++cnt;
            if (dup != 0 || cnt + v.num_empty_rows != m) {
                fprintf(stderr, "layout, kind %d: %lld fragments accounted for of %lld (%lld twice, %lld empty)\n", kind, (long long)cnt,
                        (long long)m, (long long)dup, (long long)v.num_empty_rows);
                ++bad;
            }
            polee_debug_psell_free(h);
        }
        std::vector<int32_t> par((size_t)(2 * n - 1)), js((size_t)(2 * n - 1)), par2(par), js2(js);
        if (polee_hclust_parallel(X.m, X.n, X.colptr.data(), 8, X.rowval.data(), par.data(), js.data()) != POLEE_OK) {
            fprintf(stderr, "hclust_parallel, kind %d: %s\n", kind, polee_last_error(nullptr));
            ++bad;
        } else {
            bad += check_tree("hclust_parallel", n, par, js);
        }
        if (kind < 2) {  // (the exact mode is sequential: once per structure is enough)
            if (polee_hclust(X.m, X.n, X.colptr.data(), 8, X.rowval.data(), par2.data(), js2.data()) != POLEE_OK) {
                fprintf(stderr, "hclust, kind %d: %s\n", kind, polee_last_error(nullptr));
                ++bad;
            } else {
                bad += check_tree("hclust", n, par2, js2);
            }
        }
        polee_host_cache_trim();
    }
    printf("host builders under the sanitizer: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
