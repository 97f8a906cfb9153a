// Tree construction for the Polya tree transform: the reference's greedy read-set clustering
// (src/hclust.jl:193-319) and DFS node ordering (src/hclust.jl:361-389), host side (the reference runs it on the
// CPU too).  Output = the two arrays the prep HDF5 stores (node_parent_idxs, node_js), i.e.
// exactly what polee_ptt_create takes.
//
// Determinism: the reference's result depends on tie-breaking inside DataStructures.jl's binary heaps, on the
// stability of sortperm and on Dict iteration order (remaining components).  This restatement uses the same
// heap algorithm (push = append + percolate up, pop = move last to root + percolate down, strict comparisons),
// a stable sort, neighbour lists in insertion order, and visits the remaining components in ascending node id
// (the one place where Julia's hash order cannot be followed).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <string>
#include <unordered_set>
#include <vector>

#include "common.hpp"

namespace polee {
namespace {

struct Edge {
    uint32_t j1, j2;
    float similarity;
};
struct NodeWithSize {
    uint32_t j, size;
};

// DataStructures.jl BinaryHeap: `before(a, b)` = a must be nearer the root than b (strict)
template <class T, class Before>
struct BinHeap {
    std::vector<T> xs;
    Before before;
    bool empty() const { return xs.empty(); }
    size_t size() const { return xs.size(); }
    void push(const T &x)
    {
        xs.push_back(x);
        size_t i = xs.size();  // 1-based position
        while (i > 1) {
            const size_t j = i / 2;
            if (before(x, xs[j - 1])) {
                xs[i - 1] = xs[j - 1];
                i = j;
            } else
                break;
        }
        xs[i - 1] = x;
    }
    T pop()
    {
        T top = xs[0];
        T y = xs.back();
        xs.pop_back();
        const size_t len = xs.size();
        if (len > 0) {
            size_t i = 1;
            for (;;) {
                const size_t l = 2 * i, r = l + 1;
                if (l > len) break;
                // (the four grandchildren are adjacent: their line is on its way while the children are compared -- with
                // millions of edges every level below the first few is a cache miss)
                if (4 * i <= len) __builtin_prefetch(&xs[4 * i - 1]);
                const size_t j = (r > len || before(xs[l - 1], xs[r - 1])) ? l : r;
                if (before(xs[j - 1], y)) {
                    xs[i - 1] = xs[j - 1];
                    i = j;
                } else
                    break;
            }
            xs[i - 1] = y;
        }
        return top;
    }
};
struct EdgeBefore {
    bool operator()(const Edge &a, const Edge &b) const { return a.similarity > b.similarity; }  // max-heap
};
struct SizeBefore {
    bool operator()(const NodeWithSize &a, const NodeWithSize &b) const { return a.size < b.size; }  // min-heap
};

typedef std::vector<uint32_t, default_init_allocator<uint32_t>> ReadSet;  // (resize does not zero-fill)

// hclust.jl:116-135
size_t intersection_size(const ReadSet &a, const ReadSet &b)
{
    if (a.empty() || b.empty() || a.front() > b.back() || a.back() < b.front()) return 0;
    // one set much smaller than the other: look its elements up (galloping) instead of walking both
    const ReadSet &sm = a.size() <= b.size() ? a : b, &lg = a.size() <= b.size() ? b : a;
    if (sm.size() * 24 < lg.size()) {
        size_t c = 0;
        auto from = lg.begin();
        for (uint32_t v : sm) {
            from = std::lower_bound(from, lg.end(), v);
            if (from == lg.end()) break;
            if (*from == v) ++c;
        }
        return c;
    }
    size_t i = 0, j = 0, c = 0;
    while (i < a.size() && j < b.size()) {
        if (a[i] < b[j])
            ++i;
        else if (a[i] > b[j])
            ++j;
        else {
            ++i;
            ++j;
            ++c;
        }
    }
    return c;
}
// hclust.jl:143-152 (Float64 quotient, stored as Float32 in the edge)
double relative_intersection(const ReadSet &a, const ReadSet &b)
{
    if (a.empty() && b.empty()) return 0.0;
    const size_t is = intersection_size(a, b);
    return (double)is / (double)(a.size() + b.size() - is);
}
// hclust.jl:150-190
ReadSet merge_sets(const ReadSet &a, const ReadSet &b)
{
    ReadSet out(a.size() + b.size());
    out.resize((size_t)(std::set_union(a.begin(), a.end(), b.begin(), b.end(), out.begin()) - out.begin()));
    return out;
}

struct TreeNode {
    uint32_t j = 0;           // transcript (1-based), 0 = internal
    int32_t left = -1, right = -1;
};

}  // namespace

// Returns "" or an error message.  colptr/rowval: X in CSC, 1-based, rows ascending within a column.
std::string hclust_build(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                         int32_t *node_parent_idxs, int32_t *node_js)
{
    if (n < 1 || m < 0 || !colptr || !node_parent_idxs || !node_js) return "bad argument";
    if (colptr_bytes != 4 && colptr_bytes != 8) return "colptr_bytes must be 4 or 8";
    auto cp = [&](int64_t j) -> uint64_t {
        return colptr_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(colptr)[j]
                                 : reinterpret_cast<const uint64_t *>(colptr)[j];
    };
    if (cp(0) != 1) return "colptr[0] must be 1 (1-based)";
    const int K = 25;  // neighbours compared left and right (:200)
    // PRUNED neighbour lists (opt-in, POLEE_HCLUST_PRUNE=1): the reference
    // lists a neighbour whether it shares reads or not, and re-evaluates it at every merge.  A neighbour without a common
    // read never gets an edge -- a node that shares no read with j1 and none with j2 shares none with their union --, so it
    // is left out of the lists: nine in ten of the +-25 window are such.  Every edge the reference pushes is still pushed,
    // with the same similarity, in the same relative order; what changes is that a node listed by BOTH merged nodes (once
    // without common reads) is a candidate once instead of twice, i.e. fewer duplicate edges in the heap -- which moves
    // heap positions and therefore the reference's tie-breaking among EQUAL similarities.  The exact mode reproduces the
    // reference's heap order node for node (tests/test_layouts.py against oracle/hclust_ref.py); the pruned mode builds a
    // tree by the same greedy rule with different tie-breaks.
    // (measured: 0.92 -> 0.79 s at n = 20 000, m = 3 M on 8 host threads -- the heap, not the lists, is the bulk -- so the
    // exact mode stays the default and pruning is opt-in)
    const bool prune = getenv("POLEE_HCLUST_PRUNE") != nullptr;
    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[hclust] %-24s %.3f s\n", what, now() - t_prev);
        t_prev = now();
    };

    // order transcripts by the median compatible read (:204-212); sortperm is stable
    std::vector<uint32_t> med((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        if (cp(j + 1) < cp(j)) return "colptr is not monotone";
        if (cp(j) == cp(j + 1))
            med[(size_t)j] = 0;
        else
            med[(size_t)j] = rowval[(cp(j) + cp(j + 1)) / 2 - 1];  // 1-based position div(a + b, 2)
    }
    std::vector<uint32_t> idxs((size_t)n);
    for (int64_t j = 0; j < n; ++j) idxs[(size_t)j] = (uint32_t)j;
    std::stable_sort(idxs.begin(), idxs.end(), [&](uint32_t a, uint32_t b) { return med[a] < med[b]; });

    // nodes 1..n are the leaves in that order; internal nodes are appended (:215-222)
    std::vector<TreeNode> nodes((size_t)n + 1);  // [0] unused: ids are 1-based like the reference's keys
    std::vector<ReadSet> read_sets((size_t)n + 1);
    std::vector<char> alive((size_t)n + 1, 1), deleted((size_t)n + 1, 0);
    nodes.reserve(2 * (size_t)n);
    read_sets.reserve(2 * (size_t)n);
    {
        std::atomic<int> bad{0};
        parallel_chunks((size_t)n, 4096, [&](size_t lo, size_t hi, unsigned) {
            for (size_t q = lo; q < hi; ++q) {
                const size_t j = q + 1;
                const uint32_t t = idxs[q];
                nodes[j].j = t + 1;
                read_sets[j].assign(rowval + (cp(t) - 1), rowval + (cp(t + 1) - 1));
                for (size_t e = 1; e < read_sets[j].size(); ++e)
                    if (read_sets[j][e] <= read_sets[j][e - 1]) bad = 1;
                if (!read_sets[j].empty() && (read_sets[j].front() < 1 || (int64_t)read_sets[j].back() > m)) bad = 2;
            }
        });
        if (bad == 1) return "row indexes of a column are not ascending";
        if (bad == 2) return "row index out of range";
    }

    // initial edges (:225-236)
    BinHeap<Edge, EdgeBefore> queue;
    std::vector<std::vector<uint32_t>> neighbors((size_t)n + 1);
    neighbors.reserve(2 * (size_t)n);
    lap("read sets");
    // (the n x 25 similarities are independent: computed by a few threads, then pushed in the reference's order)
    std::vector<float> sims((size_t)n * K, 0.0f);
    {
        const unsigned hw = host_threads();
        std::vector<std::thread> pool;
        for (unsigned th = 0; th < hw; ++th)
            pool.emplace_back([&, th]() {
                for (int64_t j1 = 1 + th; j1 <= n; j1 += hw)
                    for (int64_t j2 = j1 + 1; j2 <= std::min<int64_t>(j1 + K, n); ++j2)
                        sims[(size_t)(j1 - 1) * K + (size_t)(j2 - j1 - 1)] =
                            (float)relative_intersection(read_sets[(size_t)j1], read_sets[(size_t)j2]);
            });
        for (auto &t : pool) t.join();
    }
    lap("initial similarities");
    for (int64_t j1 = 1; j1 <= n; ++j1)
        for (int64_t j2 = j1 + 1; j2 <= std::min<int64_t>(j1 + K, n); ++j2) {
            const float sim = sims[(size_t)(j1 - 1) * K + (size_t)(j2 - j1 - 1)];
            if (sim > 0) queue.push(Edge{(uint32_t)j1, (uint32_t)j2, sim});
            if (sim > 0 || !prune) {
                neighbors[(size_t)j1].push_back((uint32_t)j2);
                neighbors[(size_t)j2].push_back((uint32_t)j1);
            }
        }
    std::vector<float>().swap(sims);
    lap("initial edges");

    // greedy joining (:262-308)
    std::vector<uint32_t> cand, uniq, cslot;
    std::vector<double> csim, usim;
    std::vector<uint32_t> stamp(2 * (size_t)n + 2, 0), slot(2 * (size_t)n + 2, 0);  // (node ids are 1 .. 2 n - 1; k > n >= 1 is never 0)
    std::vector<uint64_t> bits(((size_t)m >> 6) + 2, 0);  // (read ids are 1 .. m)
    const bool no_bitmap = getenv("POLEE_HCLUST_NO_BITMAP") != nullptr;  // (A/B check of the bitmap evaluation)
    const unsigned hw_threads = std::min(32u, host_threads());
    double t_merge = 0, t_eval = 0, t_eval_heavy = 0, t_pop = 0;
    size_t n_heavy = 0, n_pops = 0, n_cand = 0, n_uniq = 0;
    while (!queue.empty()) {
        const double tp0 = timing ? now() : 0.0;
        const Edge e = queue.pop();
        ++n_pops;
        if (timing) t_pop += now() - tp0;
        if (deleted[e.j1] || deleted[e.j2]) continue;  // stale edge
        const uint32_t k = (uint32_t)nodes.size();
        nodes.push_back(TreeNode{0, (int32_t)e.j1, (int32_t)e.j2});
        const double tm0 = timing ? now() : 0.0;
        read_sets.push_back(merge_sets(read_sets[e.j1], read_sets[e.j2]));
        if (timing) t_merge += now() - tm0;
        neighbors.emplace_back();
        alive.push_back(1);
        deleted.push_back(0);
        ReadSet().swap(read_sets[e.j1]);
        ReadSet().swap(read_sets[e.j2]);
        alive[e.j1] = alive[e.j2] = 0;
        deleted[e.j1] = deleted[e.j2] = 1;
        // candidates in the reference's order (:291-305): live neighbours of j1 then of j2, duplicates included
        cand.clear();
        const uint32_t pair[2][2] = {{e.j1, e.j2}, {e.j2, e.j1}};
        for (const auto &ab : pair)
            for (uint32_t l : neighbors[ab[0]])
                if (l != ab[1] && !deleted[l]) cand.push_back(l);
        // every distinct neighbour is evaluated once (the two lists overlap and contain repeats): distinct ids in the
        // order of their first appearance, found with a stamp per node instead of a sort
        uniq.clear();
        cslot.resize(cand.size());
        for (size_t c = 0; c < cand.size(); ++c) {
            const uint32_t l = cand[c];
            if (stamp[l] != k) {
                stamp[l] = k;
                slot[l] = (uint32_t)uniq.size();
                uniq.push_back(l);
            }
            cslot[c] = slot[l];
        }
        usim.resize(uniq.size());
        size_t work = 0;
        for (uint32_t l : uniq) work += read_sets[l].size() + read_sets[k].size();
        // the new node's read set is one operand of every evaluation of this merge: mark its reads in a bitmap once, a
        // neighbour's intersection is then a sum of bits over ITS reads (|neighbour| look-ups instead of a walk over
        // |neighbour| + |new node| elements; the counts, hence the similarities, are the same)
        const ReadSet &ks = read_sets[k];
        const bool use_bits = uniq.size() >= 3 && !ks.empty() && !no_bitmap;
        if (use_bits)
            for (uint32_t v : ks) bits[v >> 6] |= 1ull << (v & 63u);
        auto eval = [&](size_t lo, size_t hi) {
            for (size_t c = lo; c < hi; ++c) {
                const ReadSet &a = read_sets[uniq[c]];
                if (!use_bits) {
                    usim[c] = relative_intersection(a, ks);
                    continue;
                }
                size_t is = 0;
                if (!a.empty() && a.front() <= ks.back() && a.back() >= ks.front())
                    for (uint32_t v : a) is += (size_t)((bits[v >> 6] >> (v & 63u)) & 1ull);
                usim[c] = a.empty() ? 0.0 : (double)is / (double)(a.size() + ks.size() - is);
            }
        };
        const double te0 = timing ? now() : 0.0;
        n_cand += cand.size();
        n_uniq += uniq.size();
        const bool heavy = work > (size_t)4000000 && uniq.size() >= 8 && hw_threads > 1;
        if (heavy) ++n_heavy;
        if (heavy) {  // a heavy merge: share it out
            const unsigned nt = (unsigned)std::min<size_t>(hw_threads, uniq.size() / 4);
            std::vector<std::thread> pool;
            for (unsigned th = 1; th < nt; ++th)
                pool.emplace_back(eval, uniq.size() * th / nt, uniq.size() * (th + 1) / nt);
            eval(0, uniq.size() / nt);
            for (auto &t : pool) t.join();
        } else {
            eval(0, uniq.size());
        }
        if (use_bits)
            for (uint32_t v : ks) bits[v >> 6] = 0;
        if (timing) (heavy ? t_eval_heavy : t_eval) += now() - te0;
        csim.resize(cand.size());
        for (size_t c = 0; c < cand.size(); ++c) csim[c] = usim[cslot[c]];
        neighbors[k].reserve(cand.size());
        for (size_t c = 0; c < cand.size(); ++c) {
            const uint32_t l = cand[c];
            if (csim[c] != 0) queue.push(Edge{l, k, (float)csim[c]});
            if (csim[c] != 0 || !prune) {
                neighbors[l].push_back(k);
                neighbors[k].push_back(l);
            }
        }
    }

    if (timing)
        fprintf(stderr, "[hclust]   pops %zu (%.3f s), merge_sets %.3f s, similarities %.3f s light + %.3f s in %zu heavy merges, "
                        "candidates %zu, distinct %zu\n", n_pops, t_pop, t_merge, t_eval, t_eval_heavy, n_heavy, n_cand, n_uniq);
    lap("greedy joining");
    // remaining components: smallest first (:244-258)
    BinHeap<NodeWithSize, SizeBefore> rest;
    for (uint32_t j = 1; j < nodes.size(); ++j)
        if (alive[j]) rest.push(NodeWithSize{j, (uint32_t)(1 + read_sets[j].size())});
    while (rest.size() > 1) {
        const NodeWithSize a = rest.pop(), b = rest.pop();
        const uint32_t k = (uint32_t)nodes.size();
        nodes.push_back(TreeNode{0, (int32_t)a.j, (int32_t)b.j});
        rest.push(NodeWithSize{k, a.size + b.size});
    }
    if (rest.size() != 1) return "internal error: no root";
    const uint32_t root = rest.pop().j;
    if (nodes.size() != 2 * (size_t)n) return "internal error: node count";

    // order_nodes (:361-389): DFS, left pushed first so that the right child is visited first
    std::vector<uint32_t> stack{root};
    std::vector<int32_t> parent_of(nodes.size(), 0);
    int64_t pos = 0;
    while (!stack.empty()) {
        const uint32_t v = stack.back();
        stack.pop_back();
        node_parent_idxs[pos] = parent_of[v];
        node_js[pos] = (int32_t)nodes[v].j;
        ++pos;
        if (nodes[v].j == 0) {
            parent_of[(size_t)nodes[v].left] = (int32_t)pos;
            parent_of[(size_t)nodes[v].right] = (int32_t)pos;
            stack.push_back((uint32_t)nodes[v].left);
            stack.push_back((uint32_t)nodes[v].right);
        }
    }
    if (pos != 2 * n - 1) return "internal error: tree size";
    return "";
}

}  // namespace polee

extern "C" polee_status polee_hclust(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                     int32_t *node_parent_idxs, int32_t *node_js)
{
    const std::string err = polee::hclust_build(m, n, colptr, colptr_bytes, rowval, node_parent_idxs, node_js);
    if (!err.empty()) return polee::fail(nullptr, POLEE_ERR_BAD_ARG, "hclust: %s", err.c_str());
    return POLEE_OK;
}
