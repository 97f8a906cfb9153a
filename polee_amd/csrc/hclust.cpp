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
#include <cstring>
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
    BVec<T> xs;  // (millions of edges visited at random: on huge pages the percolate-down misses the cache, not the TLB too)
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
// a read set that lives somewhere else (a column of X, a slice of a round's arena): the parallel variant's sets
struct SetRef {
    const uint32_t *p = nullptr;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    uint32_t front() const { return p[0]; }
    uint32_t back() const { return p[n - 1]; }
    const uint32_t *begin() const { return p; }
    const uint32_t *end() const { return p + n; }
    const uint32_t *data() const { return p; }
    uint32_t operator[](size_t i) const { return p[i]; }
};

// hclust.jl:116-135
template <class Set>
size_t intersection_size(const Set &a, const Set &b)
{
    if (a.empty() || b.empty() || a.front() > b.back() || a.back() < b.front()) return 0;
    // one set much smaller than the other: look its elements up (galloping) instead of walking both
    const Set &sm = a.size() <= b.size() ? a : b, &lg = a.size() <= b.size() ? b : a;
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
template <class Set>
double relative_intersection(const Set &a, const Set &b)
{
    if (a.empty() && b.empty()) return 0.0;
    const size_t is = intersection_size(a, b);
    return (double)is / (double)(a.size() + b.size() - is);
}
// the same quotient with the reads of the other set (ks_size of them) marked in `bits`
template <class Set>
inline double bitmap_similarity(const Set &a, size_t ks_size, const std::vector<uint64_t> &bits)
{
    if (a.empty()) return 0.0;
    size_t is = 0;
    for (uint32_t v : a) is += (size_t)((bits[v >> 6] >> (v & 63u)) & 1ull);
    return (double)is / (double)(a.size() + ks_size - is);
}
// hclust.jl:150-190
ReadSet merge_sets(const ReadSet &a, const ReadSet &b)
{
    ReadSet out(a.size() + b.size());
    out.resize((size_t)(std::set_union(a.begin(), a.end(), b.begin(), b.end(), out.begin()) - out.begin()));
    return out;
}

// Large sets (the top of the tree: millions of reads per node, one or two merges per round) are cut into disjoint value
// ranges at evenly spaced elements of the larger set; the ranges are independent for counting and for merging.
struct RangeCut {
    std::vector<size_t> pa, pb;  // part p covers a[pa[p] .. pa[p+1]) and b[pb[p] .. pb[p+1])
    size_t parts() const { return pa.size() - 1; }
};
template <class Set>
RangeCut cut_ranges(const Set &a, const Set &b, size_t parts)
{
    RangeCut c;
    const Set &big = a.size() >= b.size() ? a : b;
    c.pa.push_back(0);
    c.pb.push_back(0);
    for (size_t p = 1; p < parts; ++p) {
        const uint32_t v = big[big.size() * p / parts];
        c.pa.push_back((size_t)(std::lower_bound(a.begin(), a.end(), v) - a.begin()));
        c.pb.push_back((size_t)(std::lower_bound(b.begin(), b.end(), v) - b.begin()));
    }
    c.pa.push_back(a.size());
    c.pb.push_back(b.size());
    return c;
}
inline size_t intersection_size_range(const uint32_t *a, const uint32_t *ae, const uint32_t *b, const uint32_t *be)
{
    size_t c = 0;
    while (a < ae && b < be) {
        const uint32_t x = *a, y = *b;
        c += x == y;
        a += x <= y;
        b += y <= x;
    }
    return c;
}
// |a n b| on all host threads (the counts are integers: the same number whatever the cut)
template <class Set>
size_t intersection_size_parallel(const Set &a, const Set &b, size_t grain = (size_t)1 << 20)
{
    if (a.empty() || b.empty() || a.front() > b.back() || a.back() < b.front()) return 0;
    const size_t parts = std::min<size_t>(4 * host_threads(), (a.size() + b.size()) / grain + 1);
    if (parts <= 1) return intersection_size(a, b);
    const RangeCut c = cut_ranges(a, b, parts);
    std::vector<size_t> cnt(parts, 0);
    parallel_chunks(parts, 1, [&](size_t lo, size_t hi, unsigned) {
        for (size_t p = lo; p < hi; ++p)
            cnt[p] = intersection_size_range(a.data() + c.pa[p], a.data() + c.pa[p + 1], b.data() + c.pb[p], b.data() + c.pb[p + 1]);
    });
    size_t t = 0;
    for (size_t v : cnt) t += v;
    return t;
}
// [a, ae) u [b, be) into `out` (room for both); returns the size of the union.  The loop body has no data-dependent
// branch (std::set_union's three-way branch mispredicts on every other element of interleaved sets).
inline size_t union_range(const uint32_t *a, const uint32_t *ae, const uint32_t *b, const uint32_t *be, uint32_t *out)
{
    uint32_t *o = out;
    while (a < ae && b < be) {
        const uint32_t x = *a, y = *b;
        *o++ = x < y ? x : y;
        a += x <= y;
        b += y <= x;
    }
    if (a < ae) {
        std::memcpy(o, a, (size_t)(ae - a) * sizeof(uint32_t));
        o += ae - a;
    }
    if (b < be) {
        std::memcpy(o, b, (size_t)(be - b) * sizeof(uint32_t));
        o += be - b;
    }
    return (size_t)(o - out);
}
// a u b into `out` (room for |a| + |b|); returns |a u b|
template <class Set>
size_t merge_into(const Set &a, const Set &b, uint32_t *out)
{
    return union_range(a.data(), a.data() + a.size(), b.data(), b.data() + b.size(), out);
}
// the same on all host threads: count per range, then every range merges into its place
template <class Set>
size_t merge_into_parallel(const Set &a, const Set &b, uint32_t *out, size_t grain = (size_t)1 << 20)
{
    const size_t parts = std::min<size_t>(4 * host_threads(), (a.size() + b.size()) / grain + 1);
    if (parts <= 1 || a.empty() || b.empty()) return merge_into(a, b, out);
    const RangeCut c = cut_ranges(a, b, parts);
    std::vector<size_t> off(parts + 1, 0);
    parallel_chunks(parts, 1, [&](size_t lo, size_t hi, unsigned) {
        for (size_t p = lo; p < hi; ++p)
            off[p + 1] = (c.pa[p + 1] - c.pa[p]) + (c.pb[p + 1] - c.pb[p]) -
                         intersection_size_range(a.data() + c.pa[p], a.data() + c.pa[p + 1], b.data() + c.pb[p], b.data() + c.pb[p + 1]);
    });
    for (size_t p = 0; p < parts; ++p) off[p + 1] += off[p];
    parallel_chunks(parts, 1, [&](size_t lo, size_t hi, unsigned) {
        for (size_t p = lo; p < hi; ++p)
            (void)union_range(a.data() + c.pa[p], a.data() + c.pa[p + 1], b.data() + c.pb[p], b.data() + c.pb[p + 1], out + off[p]);
    });
    return off[parts];
}

struct TreeNode {
    uint32_t j = 0;           // transcript (1-based), 0 = internal
    int32_t left = -1, right = -1;
};

// What both variants start from: the leaves in the order of their median compatible read (:204-222), idxs[j - 1] = the
// transcript (0-based) of node j; the columns of X are validated on the way.
template <class ColPtr>
std::string leaf_order(int64_t m, int64_t n, ColPtr cp, const uint32_t *rowval, std::vector<uint32_t> &idxs)
{
    // order transcripts by the median compatible read (:204-212); sortperm is stable
    std::vector<uint32_t> med((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        if (cp(j + 1) < cp(j)) return "colptr is not monotone";
        if (cp(j) == cp(j + 1))
            med[(size_t)j] = 0;
        else
            med[(size_t)j] = rowval[(cp(j) + cp(j + 1)) / 2 - 1];  // 1-based position div(a + b, 2)
    }
    idxs.resize((size_t)n);
    for (int64_t j = 0; j < n; ++j) idxs[(size_t)j] = (uint32_t)j;
    std::stable_sort(idxs.begin(), idxs.end(), [&](uint32_t a, uint32_t b) { return med[a] < med[b]; });
    std::atomic<int> bad{0};
    parallel_chunks((size_t)n, 4096, [&](size_t lo, size_t hi, unsigned) {
        for (size_t t = lo; t < hi; ++t) {
            const uint32_t *b = rowval + (cp((int64_t)t) - 1), *e = rowval + (cp((int64_t)t + 1) - 1);
            for (const uint32_t *q = b + 1; q < e; ++q)
                if (*q <= q[-1]) bad = 1;
            if (b < e && (*b < 1 || (int64_t)e[-1] > m)) bad = 2;
        }
    });
    if (bad == 1) return "row indexes of a column are not ascending";
    if (bad == 2) return "row index out of range";
    return "";
}
// the similarities of every leaf to its K successors (:225-236), sims[(j1 - 1) K + (j2 - j1 - 1)]; set(j) = node j's reads
template <class GetSet>
void leaf_similarities(int64_t m, int64_t n, int K, GetSet set, std::vector<float> &sims)
{
    // (the n x K similarities are independent: computed by a few threads; a leaf's reads are marked in the thread's bitmap
    // once, each of its K successors is then |successor| look-ups instead of a branchy walk over both sets -- the counts,
    // hence the similarities, are the same)
    sims.assign((size_t)n * K, 0.0f);
    std::vector<std::vector<uint64_t>> bitmaps(host_threads());
    parallel_chunks((size_t)n, 256, [&](size_t lo, size_t hi, unsigned th) {
        std::vector<uint64_t> &bits = bitmaps[th];
        if (bits.empty()) bits.assign(((size_t)m >> 6) + 2, 0);
        for (int64_t j1 = (int64_t)lo + 1; j1 <= (int64_t)hi; ++j1) {
            const auto &a = set((size_t)j1);
            if (a.empty()) continue;
            bool marked = false;
            for (int64_t j2 = j1 + 1; j2 <= std::min<int64_t>(j1 + K, n); ++j2) {
                const auto &b = set((size_t)j2);
                if (b.empty() || a.front() > b.back() || a.back() < b.front()) continue;
                if (!marked) {
                    for (uint32_t v : a) bits[v >> 6] |= 1ull << (v & 63u);
                    marked = true;
                }
                sims[(size_t)(j1 - 1) * K + (size_t)(j2 - j1 - 1)] = (float)bitmap_similarity(b, a.size(), bits);
            }
            if (marked)
                for (uint32_t v : a) bits[v >> 6] = 0;
        }
    });
}
// the exact mode's leaves: nodes 1..n with their own copies of the read sets (:215-222)
template <class ColPtr>
std::string leaf_setup(int64_t m, int64_t n, ColPtr cp, const uint32_t *rowval, int K, std::vector<TreeNode> &nodes,
                       std::vector<ReadSet> &read_sets, std::vector<float> &sims)
{
    std::vector<uint32_t> idxs;
    const std::string err = leaf_order(m, n, cp, rowval, idxs);
    if (!err.empty()) return err;
    nodes.assign((size_t)n + 1, TreeNode());  // [0] unused: ids are 1-based like the reference's keys
    read_sets.assign((size_t)n + 1, ReadSet());
    nodes.reserve(2 * (size_t)n);
    read_sets.reserve(2 * (size_t)n);
    parallel_chunks((size_t)n, 4096, [&](size_t lo, size_t hi, unsigned) {
        for (size_t q = lo; q < hi; ++q) {
            const uint32_t t = idxs[q];
            nodes[q + 1].j = t + 1;
            read_sets[q + 1].assign(rowval + (cp(t) - 1), rowval + (cp(t + 1) - 1));
        }
    });
    leaf_similarities(m, n, K, [&](size_t j) -> const ReadSet & { return read_sets[j]; }, sims);
    return "";
}

// The stages both variants share: the components left without a common read are joined smallest first (:244-258), then
// order_nodes (:361-389).  set_size(j) = number of reads of live node j.
template <class SetSize>
std::string finish_tree(int64_t n, std::vector<TreeNode> &nodes, const std::vector<char> &alive, SetSize set_size,
                        int32_t *node_parent_idxs, int32_t *node_js)
{
    BinHeap<NodeWithSize, SizeBefore> rest;
    for (uint32_t j = 1; j < nodes.size(); ++j)
        if (alive[j]) rest.push(NodeWithSize{j, (uint32_t)(1 + set_size(j))});
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

    std::vector<TreeNode> nodes;
    std::vector<ReadSet> read_sets;
    std::vector<float> sims;
    {
        const std::string err = leaf_setup(m, n, cp, rowval, K, nodes, read_sets, sims);
        if (!err.empty()) return err;
    }
    std::vector<char> alive((size_t)n + 1, 1), deleted((size_t)n + 1, 0);
    lap("read sets + similarities");

    // initial edges (:225-236)
    BinHeap<Edge, EdgeBefore> queue;
    std::vector<std::vector<uint32_t>> neighbors((size_t)n + 1);
    neighbors.reserve(2 * (size_t)n);
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
    return finish_tree(n, nodes, alive, [&](uint32_t j) { return read_sets[j].size(); }, node_parent_idxs, node_js);
}

// ---- the parallel variant: rounds of mutually-best merges ---------------------------------------------------------------
// The reference's greedy joining pops ONE edge at a time from a heap of millions (5.3 M pops at BASELINE's C2, every one a
// cache-missing percolate-down) and its result depends on heap positions among equal similarities -- node-for-node
// equality forbids any reordering, so the exact mode above is sequential by construction (3.8 s at C2, 20x the fit it
// feeds).  This variant keeps the rule -- join the subtrees that share the most reads first, similarities of a merged
// node recomputed against the neighbours of both halves (:262-308), same +-25 starting window, same Jaccard similarity,
// same smallest-first joining of what is left -- and replaces the global order by a local one:
//   * every edge has a priority (similarity, then a hash of its endpoints, then the endpoints): a TOTAL order;
//   * a round merges every edge that is the best edge of BOTH its endpoints (the globally best edge always is, so every
//     round makes progress; the hash breaks the long chains that equal similarities -- small rationals: the rule, not
//     the exception -- would form under an id tie-break);
//   * the merges of a round are independent: unions, candidate lists and similarities run on all host threads;
//   * new nodes are numbered in the order of their edges' priorities: the tree does not depend on the thread count.
// Neighbours without a common read are not listed (a node sharing no read with either half shares none with the union;
// it can only become similar again through a LATER merge of its own neighbours, which lists it then).
// Where it differs from the exact mode: an edge is merged as soon as nothing better touches either endpoint, although a
// better edge may exist elsewhere (irrelevant for the tree: disjoint subtrees) or may APPEAR at an endpoint later through
// a neighbour's merge (Jaccard similarity is not reducible, so this changes some joins).  The tree is a heuristic
// parameterisation either way (the reference ships two others, polee.jl: random and sequential trees); validated by the
// fit it gives (tests/test_gpu_hclust.py: E[lp] of the fit on the parallel tree against the exact tree's).
namespace {

struct Nbr {
    uint32_t id;
    float sim;
};
struct EdgePri {  // larger = merged first
    uint64_t key;  // similarity bits (positive floats order like their bit patterns) << 32 | hash
    uint32_t lo, hi;
    bool operator<(const EdgePri &o) const
    {
        if (key != o.key) return key < o.key;
        if (lo != o.lo) return lo > o.lo;
        return hi > o.hi;
    }
    bool operator==(const EdgePri &o) const { return lo == o.lo && hi == o.hi; }
};
inline EdgePri edge_pri(uint32_t a, uint32_t b, float sim)
{
    const uint32_t lo = std::min(a, b), hi = std::max(a, b);
    uint64_t h = (uint64_t)lo * 0x9E3779B97F4A7C15ull ^ ((uint64_t)hi * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull);
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    uint32_t sb;
    memcpy(&sb, &sim, 4);
    return EdgePri{((uint64_t)sb << 32) | (uint32_t)h, lo, hi};
}

}  // namespace

std::string hclust_build_rounds(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                int32_t *node_parent_idxs, int32_t *node_js)
{
    if (n < 1 || m < 0 || !colptr || !node_parent_idxs || !node_js) return "bad argument";
    if (colptr_bytes != 4 && colptr_bytes != 8) return "colptr_bytes must be 4 or 8";
    auto cp = [&](int64_t j) -> uint64_t {
        return colptr_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(colptr)[j]
                                 : reinterpret_cast<const uint64_t *>(colptr)[j];
    };
    if (cp(0) != 1) return "colptr[0] must be 1 (1-based)";
    const int K = 25;
    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[hclust/rounds] %-24s %.3f s\n", what, now() - t_prev);
        t_prev = now();
    };
    // Read sets are views: a leaf's is its column of X where it lies; a merged node's is a slice of its round's arena (one
    // block per round, room for |a| + |b| per merge, from the block cache: no allocation per merge, and the next round --
    // and the next sample -- write into memory that is already resident).  An arena goes back when its last set retires.
    const size_t cap = 2 * (size_t)n + 1;
    std::vector<TreeNode> nodes(cap);
    std::vector<SetRef> read_sets(cap);
    std::vector<int32_t> arena_of(cap, -1);
    std::vector<RawVec<uint32_t>> arenas;
    std::vector<int> arena_live;
    std::vector<float> sims;
    {
        std::vector<uint32_t> idxs;
        const std::string err = leaf_order(m, n, cp, rowval, idxs);
        if (!err.empty()) return err;
        for (int64_t q = 0; q < n; ++q) {
            const uint32_t t = idxs[(size_t)q];
            nodes[(size_t)q + 1].j = t + 1;
            read_sets[(size_t)q + 1] = SetRef{rowval + (cp(t) - 1), (size_t)(cp(t + 1) - cp(t))};
        }
        leaf_similarities(m, n, K, [&](size_t j) -> const SetRef & { return read_sets[j]; }, sims);
    }
    lap("leaves + similarities");
    std::vector<std::vector<Nbr>> adj(cap);
    std::vector<char> alive(cap, 0);
    std::vector<uint32_t> best(cap, 0), into(cap, 0), stamp(cap, 0);
    std::vector<uint32_t> set_size(cap, 0);
    for (int64_t j = 1; j <= n; ++j) alive[(size_t)j] = 1, set_size[(size_t)j] = (uint32_t)read_sets[(size_t)j].size();
    for (int64_t j1 = 1; j1 <= n; ++j1)
        for (int64_t j2 = j1 + 1; j2 <= std::min<int64_t>(j1 + K, n); ++j2) {
            const float sim = sims[(size_t)(j1 - 1) * K + (size_t)(j2 - j1 - 1)];
            if (sim > 0) {
                adj[(size_t)j1].push_back(Nbr{(uint32_t)j2, sim});
                adj[(size_t)j2].push_back(Nbr{(uint32_t)j1, sim});
            }
        }
    std::vector<float>().swap(sims);
    std::vector<uint32_t> dirty;
    for (int64_t j = 1; j <= n; ++j)
        if (!adj[(size_t)j].empty()) dirty.push_back((uint32_t)j);
    lap("initial edges");

    uint32_t next_id = (uint32_t)n + 1;
    struct Pair {
        EdgePri pri;
        uint32_t k;
    };
    std::vector<Pair> pairs;
    std::vector<std::vector<uint32_t>> cands;  // per pair: the distinct live neighbours of both halves (new ids applied)
    std::vector<size_t> task_ptr;
    std::vector<uint32_t> next_dirty;
    std::vector<std::vector<uint32_t>> dirty_parts(host_threads());
    std::vector<std::atomic_flag> locks(4096);
    for (auto &f : locks) f.clear();
    std::vector<float> tsim;
    std::vector<size_t> slot;  // a merge's place in the round's arena
    // a merge is "heavy" when its sets are large AND its work exceeds a thread's fair share of the round: then it is cut
    // into value ranges over all threads instead of being one task (POLEE_HCLUST_HEAVY: the size threshold, for tests)
    const size_t HEAVY = getenv("POLEE_HCLUST_HEAVY") ? (size_t)atol(getenv("POLEE_HCLUST_HEAVY")) : (size_t)1 << 22;
    std::vector<char> is_heavy;
    // (a range per ~1 M elements: starting a host thread costs about what merging 50 k elements does)
    const size_t grain = std::max<size_t>(1, std::min<size_t>((size_t)1 << 20, HEAVY));
    std::vector<std::vector<uint64_t>> bitmaps(host_threads());
    size_t rounds = 0, n_eval = 0;
    double t_best = 0, t_union = 0, t_eval = 0, t_apply = 0;
    while (!dirty.empty()) {
        ++rounds;
        double t0 = timing ? now() : 0.0;
        // A: the best live edge of every node whose neighbourhood changed (dead neighbours leave the list here)
        parallel_chunks(dirty.size(), 2048, [&](size_t lo, size_t hi, unsigned) {
            for (size_t q = lo; q < hi; ++q) {
                const uint32_t a = dirty[q];
                std::vector<Nbr> &L = adj[a];
                size_t w = 0;
                EdgePri bp{0, 0, 0};
                uint32_t b = 0;
                for (size_t e = 0; e < L.size(); ++e) {
                    if (!alive[L[e].id]) continue;
                    L[w++] = L[e];
                    const EdgePri pr = edge_pri(a, L[e].id, L[e].sim);
                    if (b == 0 || bp < pr) bp = pr, b = L[e].id;
                }
                L.resize(w);
                best[a] = b;
            }
        });
        // B: the edges that are the best of both endpoints, in the order of their priorities
        pairs.clear();
        for (uint32_t a : dirty) {
            const uint32_t b = best[a];
            if (!b || best[b] != a) continue;
            float sim = 0.0f;
            for (const Nbr &e : adj[a])
                if (e.id == b) sim = e.sim;
            pairs.push_back(Pair{edge_pri(a, b, sim), 0});
        }
        std::sort(pairs.begin(), pairs.end(), [](const Pair &x, const Pair &y) { return y.pri < x.pri; });
        pairs.erase(std::unique(pairs.begin(), pairs.end(), [](const Pair &x, const Pair &y) { return x.pri == y.pri; }),
                    pairs.end());
        if (pairs.empty()) break;  // (cannot happen while an edge is left: the best edge overall is mutual)
        const uint32_t base = next_id;  // ids below are the old nodes
        for (Pair &pr : pairs) {
            pr.k = next_id++;
            into[pr.pri.lo] = into[pr.pri.hi] = pr.k;
            nodes[pr.k] = TreeNode{0, (int32_t)pr.pri.lo, (int32_t)pr.pri.hi};
        }
        if (timing) t_best += now() - t0, t0 = now();
        // C1: unions (into this round's arena) and candidate lists (independent per merge)
        cands.resize(pairs.size());
        slot.assign(pairs.size() + 1, 0);
        for (size_t q = 0; q < pairs.size(); ++q)
            slot[q + 1] = slot[q] + read_sets[pairs[q].pri.lo].size() + read_sets[pairs[q].pri.hi].size();
        const size_t round_work = slot.back();
        const int32_t arena_id = (int32_t)arenas.size();
        arenas.emplace_back();
        arenas.back().resize(round_work);
        arena_live.push_back((int)pairs.size());
        uint32_t *const arena = arenas.back().data();
        // a set retires: its arena goes back to the block cache with its last set (called from the parallel loop below:
        // only the thread that brings the count to zero touches the arena)
        auto retire = [&](uint32_t x) {
            const int32_t id = arena_of[x];
            read_sets[x] = SetRef{};
            if (id >= 0 && __atomic_sub_fetch(&arena_live[(size_t)id], 1, __ATOMIC_ACQ_REL) == 0) RawVec<uint32_t>().swap(arenas[(size_t)id]);
        };
        // (the few merges of millions of reads first, one at a time on all threads; the rest one merge per task)
        is_heavy.assign(pairs.size(), 0);
        for (size_t q = 0; q < pairs.size(); ++q) {
            const uint32_t a = pairs[q].pri.lo, b = pairs[q].pri.hi, k = pairs[q].k;
            const size_t w = slot[q + 1] - slot[q];
            if (w < HEAVY || w * host_threads() < round_work) continue;
            is_heavy[q] = 1;
            read_sets[k] = SetRef{arena + slot[q], merge_into_parallel(read_sets[a], read_sets[b], arena + slot[q], grain)};
        }
        parallel_chunks(pairs.size(), 8, [&](size_t lo, size_t hi, unsigned) {
            for (size_t q = lo; q < hi; ++q) {
                const uint32_t a = pairs[q].pri.lo, b = pairs[q].pri.hi, k = pairs[q].k;
                if (!is_heavy[q]) read_sets[k] = SetRef{arena + slot[q], merge_into(read_sets[a], read_sets[b], arena + slot[q])};
                arena_of[k] = arena_id;
                set_size[k] = (uint32_t)read_sets[k].size();
                std::vector<uint32_t> &c = cands[q];
                c.clear();
                for (uint32_t half : {a, b})
                    for (const Nbr &e : adj[half]) {
                        if (e.id == a || e.id == b) continue;
                        c.push_back(into[e.id] ? into[e.id] : e.id);
                    }
                std::sort(c.begin(), c.end());
                c.erase(std::unique(c.begin(), c.end()), c.end());
                // the halves retire here, on this thread: nobody else reads their sets or lists (other merges see them
                // through `into` only)
                retire(a);
                retire(b);
                std::vector<Nbr>().swap(adj[a]);
                std::vector<Nbr>().swap(adj[b]);
            }
        });
        if (timing) t_union += now() - t0, t0 = now();
        // C2: similarities of the new nodes to their candidates.  A new node's reads are marked in a bitmap once and
        // every candidate is |candidate| look-ups (as in the exact mode).  Light merges: one merge per task, the thread's
        // own bitmap.  Heavy merges (the top of the tree: few per round, millions of reads): one at a time, one shared
        // bitmap, the candidates spread over the threads.
        task_ptr.assign(pairs.size() + 1, 0);
        for (size_t q = 0; q < pairs.size(); ++q) task_ptr[q + 1] = task_ptr[q] + cands[q].size();
        const size_t ntasks = task_ptr.back();
        n_eval += ntasks;
        tsim.resize(ntasks);
        auto range_hit = [&](const SetRef &a, const SetRef &ks) {
            return !a.empty() && !ks.empty() && a.front() <= ks.back() && a.back() >= ks.front();
        };
        parallel_chunks(pairs.size(), 4, [&](size_t lo, size_t hi, unsigned th) {
            std::vector<uint64_t> &bits = bitmaps[th];
            for (size_t q = lo; q < hi; ++q) {
                const SetRef &ks = read_sets[pairs[q].k];
                const size_t nc = cands[q].size();
                if (is_heavy[q]) continue;  // (below)
                const bool use_bits = nc >= 3;
                if (use_bits) {
                    if (bits.empty()) bits.assign(((size_t)m >> 6) + 2, 0);
                    for (uint32_t v : ks) bits[v >> 6] |= 1ull << (v & 63u);
                }
                for (size_t c = 0; c < nc; ++c) {
                    const SetRef &a = read_sets[cands[q][c]];
                    tsim[task_ptr[q] + c] = !range_hit(a, ks) ? 0.0f
                                            : (float)(use_bits ? bitmap_similarity(a, ks.size(), bits) : relative_intersection(a, ks));
                }
                if (use_bits)
                    for (uint32_t v : ks) bits[v >> 6] = 0;
            }
        });
        for (size_t q = 0; q < pairs.size(); ++q) {
            const SetRef &ks = read_sets[pairs[q].k];
            const size_t nc = cands[q].size();
            if (!is_heavy[q]) continue;
            for (size_t c = 0; c < nc; ++c) {  // every evaluation on all threads, cut into value ranges
                const SetRef &a = read_sets[cands[q][c]];
                const size_t is = intersection_size_parallel(a, ks, grain);
                tsim[task_ptr[q] + c] = a.empty() ? 0.0f : (float)((double)is / (double)(a.size() + ks.size() - is));
            }
        }
        if (timing) t_eval += now() - t0, t0 = now();
        // D: the new nodes' lists, their entries in the old neighbours' lists, the halves retire.  In parallel over the
        // merges: an old neighbour's list (and its dirty stamp) is taken under one of 4096 striped spin locks.  The order
        // of a list's entries then depends on timing, but nothing reads a list by position: the best edge is a maximum
        // over a total order, candidate lists are sorted, similarities are looked up by neighbour id.
        for (auto &v : dirty_parts) v.clear();
        parallel_chunks(pairs.size(), 64, [&](size_t lo, size_t hi, unsigned th) {
            std::vector<uint32_t> &nd = dirty_parts[th];
            for (size_t q = lo; q < hi; ++q) {
                const uint32_t a = pairs[q].pri.lo, b = pairs[q].pri.hi, k = pairs[q].k;
                std::vector<Nbr> &L = adj[k];
                L.reserve(task_ptr[q + 1] - task_ptr[q]);
                for (size_t t = task_ptr[q]; t < task_ptr[q + 1]; ++t) {
                    const uint32_t l = cands[q][t - task_ptr[q]];
                    if (!(tsim[t] > 0)) continue;
                    L.push_back(Nbr{l, tsim[t]});
                    if (l < base) {  // an old node: it learns about k (a new one lists k itself)
                        std::atomic_flag &lk = locks[l & 4095u];
                        while (lk.test_and_set(std::memory_order_acquire)) {
                        }
                        adj[l].push_back(Nbr{k, tsim[t]});
                        const bool first = stamp[l] != (uint32_t)rounds;
                        stamp[l] = (uint32_t)rounds;
                        lk.clear(std::memory_order_release);
                        if (first) nd.push_back(l);
                    }
                }
                alive[a] = alive[b] = 0;
                alive[k] = 1;
                if (!L.empty()) nd.push_back(k);
            }
        });
        next_dirty.clear();
        for (const auto &v : dirty_parts) next_dirty.insert(next_dirty.end(), v.begin(), v.end());
        for (const Pair &pr : pairs) into[pr.pri.lo] = into[pr.pri.hi] = 0;
        // (an old node that was listed but merged in this very round is dead now; step A skips nothing for it: drop it)
        dirty.clear();
        for (uint32_t l : next_dirty)
            if (alive[l]) dirty.push_back(l);
        std::sort(dirty.begin(), dirty.end());
        if (timing) t_apply += now() - t0;
    }
    if (timing)
        fprintf(stderr, "[hclust/rounds]   %zu rounds, %zu similarity evaluations; best/pairs %.3f s, unions %.3f s, "
                        "similarities %.3f s, apply %.3f s\n", rounds, n_eval, t_best, t_union, t_eval, t_apply);
    lap("joining in rounds");
    nodes.resize(next_id);
    alive.resize(next_id);
    return finish_tree(n, nodes, alive, [&](uint32_t j) { return (size_t)set_size[j]; }, node_parent_idxs, node_js);
}

}  // namespace polee

namespace polee {
std::string hclust_finish_from_arrays(int64_t n, uint32_t num_nodes, const int32_t *left, const int32_t *right, const uint8_t *alive_in,
                                      const uint32_t *set_len, const uint32_t *leaf_transcript, int32_t *node_parent_idxs, int32_t *node_js)
{
    std::vector<TreeNode> nodes(num_nodes);
    std::vector<char> alive(num_nodes, 0);
    for (uint32_t j = 1; j < num_nodes; ++j) {
        nodes[j] = j <= (uint32_t)n ? TreeNode{leaf_transcript[j - 1] + 1, -1, -1} : TreeNode{0, left[j], right[j]};
        alive[j] = (char)alive_in[j];
    }
    return finish_tree(n, nodes, alive, [&](uint32_t j) { return (size_t)set_len[j]; }, node_parent_idxs, node_js);
}
}  // namespace polee

//  nothing may unwind through the C ABI: the builders allocate on the calling thread and inside
// parallel_chunks workers (which hand the first exception back to the caller, common.hpp)
template <class F>
static polee_status hclust_guarded(F &&f)
{
    try {
        const std::string err = f();
        if (!err.empty()) return polee::fail(nullptr, POLEE_ERR_BAD_ARG, "hclust: %s", err.c_str());
        return POLEE_OK;
    } catch (const std::bad_alloc &) {
        return polee::fail(nullptr, POLEE_ERR_OOM, "hclust: out of host memory");
    } catch (const std::exception &e) {
        return polee::fail(nullptr, POLEE_ERR_UNSUPPORTED, "hclust: %s", e.what());
    } catch (...) {
        return polee::fail(nullptr, POLEE_ERR_UNSUPPORTED, "hclust: unknown exception");
    }
}

extern "C" polee_status polee_hclust(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                     int32_t *node_parent_idxs, int32_t *node_js)
{
    return hclust_guarded([&] { return polee::hclust_build(m, n, colptr, colptr_bytes, rowval, node_parent_idxs, node_js); });
}

extern "C" polee_status polee_hclust_parallel(int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                              const uint32_t *rowval, int32_t *node_parent_idxs, int32_t *node_js)
{
    return hclust_guarded([&] { return polee::hclust_build_rounds(m, n, colptr, colptr_bytes, rowval, node_parent_idxs, node_js); });
}
