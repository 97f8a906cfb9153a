// Device-side tree construction: the ROUNDS variant of the clustering heuristic (hclust.cpp, hclust_build_rounds -- the
// reference's greedy read-set clustering, src/hclust.jl:193-319, with the global order replaced by rounds of mutually-best
// merges) as kernels.  The same tree as polee_hclust_parallel, node for node: priorities are a total order, new nodes are numbered
// in that order, similarities are quotients of integer counts (tests/test_gpu_hclust.py compares the arrays).
//
// Per round: the best live edge of every node (atomic max over the priorities), the edges that are the best of BOTH endpoints,
// sorted by priority = the round's merges; their read sets united by merge path (tiles of 1 024 merged elements, each set read once,
// consecutively), the candidates = neighbours of both halves, |new set n candidate| by binary searches
// spread over as many blocks as the smaller set needs, the new edges appended, dead edges dropped.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <string>
#include <deque>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "hclust_device.hpp"
#include "loglik_internal.hpp"  // (xbuild_device_view)

namespace polee {
namespace {

#define HD_HIP(expr) POLEE_HIP_TRY(ctx, expr)

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
template <typename K, typename V>
hipError_t sort_pairs(Scratch &tmp, const K *kin, K *kout, const V *vin, V *vout, size_t count, bool descending, hipStream_t stream)
{
    size_t bytes = 0;
    hipError_t e = descending ? rocprim::radix_sort_pairs_desc(nullptr, bytes, kin, kout, vin, vout, count, 0, 8 * sizeof(K), stream)
                              : rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, count, 0, 8 * sizeof(K), stream);
    if (e != hipSuccess) return e;
    if ((e = tmp.need(bytes)) != hipSuccess) return e;
    return descending ? rocprim::radix_sort_pairs_desc(tmp.p, bytes, kin, kout, vin, vout, count, 0, 8 * sizeof(K), stream)
                      : rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, count, 0, 8 * sizeof(K), stream);
}
struct ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

// priority of an edge (hclust.cpp, edge_pri): similarity bits << 32 | hash of the endpoints; ties: smaller lo, then smaller hi
__device__ inline uint64_t pri_key(uint32_t a, uint32_t b, float sim)
{
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    uint64_t h = (uint64_t)lo * 0x9E3779B97F4A7C15ull ^ ((uint64_t)hi * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull);
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return ((uint64_t)__float_as_uint(sim) << 32) | (uint32_t)h;
}

struct Nodes {
    const uint32_t **set_p;  // [cap] the node's reads, ascending (a column of X, or a slice of a round's arena)
    uint32_t *set_len;
    uint8_t *alive;
    uint32_t *into;  // the node this one is merged into in the current round, or 0
    unsigned long long *bestkey, *besttie;
    uint32_t *best;
    int32_t *left, *right;
};
struct Edges {
    uint32_t *src, *dst;
    float *sim;
};

// ---- leaves ----------------------------------------------------------------------------------------------------------------
__global__ void hd_median_kernel(int64_t n, int64_t m, const uint64_t *cp, const uint32_t *rowval, uint32_t *med, uint32_t *idx, uint32_t *err)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t a = cp[j], b = cp[j + 1];
    if (b < a) {
        atomicMax(err, 3u);
        med[j] = 0;
        idx[j] = (uint32_t)j;
        return;
    }
    med[j] = a == b ? 0u : rowval[(a + b) / 2 - 1];
    idx[j] = (uint32_t)j;
    if (b > a && (rowval[a - 1] < 1 || (int64_t)rowval[b - 2] > m)) atomicMax(err, 2u);
}
__global__ void hd_validate_kernel(uint64_t nnz, const uint32_t *rowval, const uint32_t *colstart_flag, uint32_t *err)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 || k >= nnz) return;
    if (!colstart_flag[k] && rowval[k] <= rowval[k - 1]) atomicMax(err, 1u);
}
__global__ void hd_colstart_kernel(int64_t n, const uint64_t *cp, uint64_t nnz, uint32_t *flag)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (cp[j + 1] > cp[j] && cp[j] - 1 < nnz) flag[cp[j] - 1] = 1;
}
__global__ void hd_leaves_kernel(int64_t n, const uint64_t *cp, const uint32_t *rowval, const uint32_t *idxs, Nodes N, uint32_t *leaf_t)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint32_t t = idxs[q];
    N.set_p[q + 1] = rowval + (cp[t] - 1);
    N.set_len[q + 1] = (uint32_t)(cp[t + 1] - cp[t]);
    N.alive[q + 1] = 1;
    leaf_t[q] = t;
}
__global__ void hd_leaf_tasks_kernel(int64_t n, int K, uint32_t *tx, uint32_t *ty)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int64_t j1 = i / K + 1, j2 = j1 + 1 + i % K;
    tx[i] = (uint32_t)j1;
    ty[i] = j2 <= n ? (uint32_t)j2 : 0u;  // (0: no such pair)
}

// ---- |X n Y| of a list of tasks ---------------------------------------------------------------------------------------------
constexpr uint32_t ISECT_CHUNK = 512;  // elements of the smaller set per WAVE (the waves of a block work on tasks of their own:
                                       // a piece of work is a chain of dependent loads -- which task, its sets, where its run can
                                       // hit -- and four times as many of them are in flight than with a block per piece)
__global__ void hd_task_blocks_kernel(uint32_t T, const uint32_t *tx, const uint32_t *ty, Nodes N, uint32_t *nblk)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    uint32_t nb = 0;
    const uint32_t x = tx[t], y = ty[t];
    if (x && y) {
        const uint32_t lx = N.set_len[x], ly = N.set_len[y];
        if (lx && ly) {
            const uint32_t *px = N.set_p[x], *py = N.set_p[y];
            if (!(px[0] > py[ly - 1] || px[lx - 1] < py[0])) nb = (min(lx, ly) + ISECT_CHUNK - 1) / ISECT_CHUNK;
        }
    }
    nblk[t] = nb;
}
// piece -> task (a task's pieces are consecutive): the task's number at its first piece, then a running maximum
__global__ void hd_piece_heads_kernel(uint32_t T, const uint32_t *nblk, const uint32_t *bscan, uint32_t *pmap)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T && nblk[t]) pmap[bscan[t]] = t;
}
// A piece = 512 consecutive elements of the smaller set, one wave.  What a piece costs is its chain of DEPENDENT loads, so the
// chain is kept short: the task from a map; the part of the larger set the piece's sorted run can hit by a 64-ary search (every
// lane probes one of 64 evenly spaced positions: two or three rounds instead of two binary searches); then every lane's eight
// elements searched in lockstep (eight independent loads per round).
__global__ __launch_bounds__(256) void hd_isect_kernel(uint32_t npieces, const uint32_t *pmap, const uint32_t *tx, const uint32_t *ty, Nodes N,
                                                       const uint32_t *bscan, uint32_t *count)
{
    const uint32_t piece = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (piece >= npieces) return;
    const uint32_t t = pmap[piece], chunk = piece - bscan[t];
    const uint32_t x = tx[t], y = ty[t];
    const uint32_t lx = N.set_len[x], ly = N.set_len[y];
    const uint32_t *ps = lx <= ly ? N.set_p[x] : N.set_p[y], *pb = lx <= ly ? N.set_p[y] : N.set_p[x];
    const uint32_t ls = min(lx, ly), lb = max(lx, ly);
    const uint32_t e0 = chunk * ISECT_CHUNK, e1 = min(ls, e0 + ISECT_CHUNK);
    const uint32_t vfirst = ps[e0], vlast = ps[e1 - 1];
    // [wlo, whi): the elements of the larger set in [vfirst, vlast].  64-ary search: the answer lies in [a, b]; the lanes probe
    // a, a + step, a + 2 step, ...: the probes that satisfy the predicate are a prefix (the set ascends)
    uint32_t wlo, whi;
    {
        uint32_t a = 0, b = lb;  // first position whose element is >= vfirst
        while (a < b) {
            const uint32_t step = (b - a + 63u) / 64u;
            const uint32_t pos = a + lane * step;
            const uint32_t nb = (uint32_t)__popcll(__ballot(pos < b && pb[pos < b ? pos : a] < vfirst));
            if (nb == 0) {
                b = a;
            } else {
                const uint32_t na = a + (nb - 1) * step + 1;
                b = min(a + nb * step, b);
                a = na;
            }
        }
        wlo = a;
        b = lb;  // first position whose element is > vlast (not before wlo)
        while (a < b) {
            const uint32_t step = (b - a + 63u) / 64u;
            const uint32_t pos = a + lane * step;
            const uint32_t nb = (uint32_t)__popcll(__ballot(pos < b && pb[pos < b ? pos : a] <= vlast));
            if (nb == 0) {
                b = a;
            } else {
                const uint32_t na = a + (nb - 1) * step + 1;
                b = min(a + nb * step, b);
                a = na;
            }
        }
        whi = a;
    }
    uint32_t c = 0;
    if (whi > wlo) {
        constexpr int E = ISECT_CHUNK / 64;
        uint32_t v[E], lo_[E], hi_[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = e0 + lane + 64u * (uint32_t)e;
            v[e] = i < e1 ? ps[i] : 0xffffffffu;
            lo_[e] = wlo;
            hi_[e] = whi;
        }
        const int rounds = 32 - __builtin_clz(whi - wlo);  // enough halvings for the widest interval
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (lo_[e] < hi_[e]) {
                    const uint32_t mid = (lo_[e] + hi_[e]) >> 1;
                    if (pb[mid] < v[e]) lo_[e] = mid + 1; else hi_[e] = mid;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = e0 + lane + 64u * (uint32_t)e;
            c += i < e1 && lo_[e] < whi && pb[lo_[e]] == v[e];
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0 && c) atomicAdd(&count[t], c);
}
// Float64 quotient stored as Float32 (hclust.jl:143-152)
__global__ void hd_sim_kernel(uint32_t T, const uint32_t *tx, const uint32_t *ty, Nodes N, const uint32_t *count, float *sim)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t x = tx[t], y = ty[t];
    float s = 0.0f;
    if (x && y) {
        const uint32_t lx = N.set_len[x], ly = N.set_len[y], is = count[t];
        if (lx && ly) s = (float)((double)is / (double)((uint64_t)lx + ly - is));
    }
    sim[t] = s;
}

// ---- edges ------------------------------------------------------------------------------------------------------------------
// new edges of a task list: x -> y always, y -> x when y is an old node (y < base); flags first, then placed by a scan
__global__ void hd_edge_count_kernel(uint32_t T, const uint32_t *ty, const float *sim, uint32_t base, uint32_t *cnt)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    cnt[t] = sim[t] > 0.0f ? (ty[t] < base ? 2u : 1u) : 0u;
}
__global__ void hd_edge_emit_kernel(uint32_t T, const uint32_t *tx, const uint32_t *ty, const float *sim, uint32_t base, const uint32_t *pos,
                                    uint64_t at, Edges E)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || !(sim[t] > 0.0f)) return;
    const uint64_t p = at + pos[t];
    E.src[p] = tx[t];
    E.dst[p] = ty[t];
    E.sim[p] = sim[t];
    if (ty[t] < base) {
        E.src[p + 1] = ty[t];
        E.dst[p + 1] = tx[t];
        E.sim[p + 1] = sim[t];
    }
}
__global__ void hd_best1_kernel(uint64_t ne, Edges E, Nodes N)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const uint32_t s = E.src[e], d = E.dst[e];
    if (!N.alive[s] || !N.alive[d]) return;
    atomicMax(&N.bestkey[s], (unsigned long long)pri_key(s, d, E.sim[e]));
}
__global__ void hd_best2_kernel(uint64_t ne, Edges E, Nodes N)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const uint32_t s = E.src[e], d = E.dst[e];
    if (!N.alive[s] || !N.alive[d]) return;
    if ((unsigned long long)pri_key(s, d, E.sim[e]) != N.bestkey[s]) return;
    const uint32_t lo = min(s, d), hi = max(s, d);
    atomicMin(&N.besttie[s], ((unsigned long long)lo << 32) | hi);
}
__global__ void hd_best3_kernel(uint32_t cap, Nodes N)
{
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= cap) return;
    uint32_t b = 0;
    if (N.alive[a] && N.bestkey[a] != 0) {
        const unsigned long long tie = N.besttie[a];
        const uint32_t lo = (uint32_t)(tie >> 32), hi = (uint32_t)tie;
        b = lo == a ? hi : lo;
    }
    N.best[a] = b;
}
__global__ void hd_mutual_kernel(uint32_t cap, Nodes N, uint32_t *flag)
{
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= cap) return;
    const uint32_t b = N.best[a];
    flag[a] = b && a < b && N.best[b] == a;
}
__global__ void hd_pairs_kernel(uint32_t cap, Nodes N, const uint32_t *flag, const uint32_t *pos, unsigned long long *pkey, unsigned long long *ptie)
{
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= cap || !flag[a]) return;
    pkey[pos[a]] = N.bestkey[a];
    ptie[pos[a]] = ((unsigned long long)a << 32) | N.best[a];
}
// the round's merges in the order of their priorities: ids, halves, room in the arena
__global__ void hd_ids_kernel(uint32_t P, const unsigned long long *ptie, uint32_t base, Nodes N, uint32_t *plo, uint32_t *phi, uint32_t *wlen,
                              uint32_t *blen)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    const uint32_t lo = (uint32_t)(ptie[q] >> 32), hi = (uint32_t)ptie[q];
    const uint32_t k = base + q;
    plo[q] = lo;
    phi[q] = hi;
    N.into[lo] = N.into[hi] = k;
    N.left[k] = (int32_t)lo;
    N.right[k] = (int32_t)hi;
    wlen[q] = N.set_len[lo] + N.set_len[hi];
    blen[q] = N.set_len[hi];
}

// ---- unions -----------------------------------------------------------------------------------------------------------------
// A = the set of the pair's lower node, B = of the other.  Element g of the concatenated B lists: is it in A, and how many
// elements of A lie below it.
__device__ inline uint32_t find_segment(const uint64_t *scan, uint32_t P, uint64_t g)  // the last q with scan[q] <= g
{
    uint32_t lo = 0, hi = P;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (scan[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}
__global__ void hd_diff_u64_kernel(uint32_t N, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = a[i] - b[i];
}
__global__ void hd_union_b_kernel(uint64_t nb, uint32_t P, const uint64_t *bscan, const uint32_t *plo, const uint32_t *phi, Nodes N,
                                  uint32_t *isdup, uint32_t *lba)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nb) return;
    const uint32_t q = find_segment(bscan, P, g);
    const uint32_t j = (uint32_t)(g - bscan[q]);
    const uint32_t *A = N.set_p[plo[q]];
    const uint32_t la = N.set_len[plo[q]];
    const uint32_t y = N.set_p[phi[q]][j];
    uint32_t a = 0, b = la;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (A[mid] < y) a = mid + 1; else b = mid;
    }
    lba[g] = a;
    isdup[g] = a < la && A[a] == y;
}
__global__ void hd_union_place_b_kernel(uint64_t nb, uint32_t P, const uint64_t *bscan, const uint32_t *phi, Nodes N, const uint32_t *isdup,
                                        const uint64_t *dscan, const uint32_t *lba, const uint64_t *slot, uint32_t *arena)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nb || isdup[g]) return;
    const uint32_t q = find_segment(bscan, P, g);
    const uint32_t j = (uint32_t)(g - bscan[q]);
    const uint64_t dup_before = dscan[g] - dscan[bscan[q]];
    arena[slot[q] + lba[g] + j - dup_before] = N.set_p[phi[q]][j];
}
__global__ void hd_union_place_a_kernel(uint64_t na, uint32_t P, const uint64_t *ascan, const uint64_t *bscan, const uint32_t *plo, const uint32_t *phi,
                                        Nodes N, const uint64_t *dscan, const uint64_t *slot, uint32_t *arena)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= na) return;
    const uint32_t q = find_segment(ascan, P, g);
    const uint32_t i = (uint32_t)(g - ascan[q]);
    const uint32_t x = N.set_p[plo[q]][i];
    const uint32_t *B = N.set_p[phi[q]];
    const uint32_t lb = N.set_len[phi[q]];
    uint32_t a = 0, b = lb;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (B[mid] < x) a = mid + 1; else b = mid;
    }
    const uint64_t dup_before = dscan[bscan[q] + a] - dscan[bscan[q]];
    arena[slot[q] + i + a - dup_before] = x;
}
__global__ void hd_union_finish_kernel(uint32_t P, uint32_t base, const uint32_t *plo, const uint32_t *phi, const uint64_t *bscan, const uint64_t *dscan,
                                       const uint64_t *slot, uint32_t *arena, Nodes N)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    const uint32_t k = base + q;
    const uint64_t dups = dscan[bscan[q + 1]] - dscan[bscan[q]];
    N.set_p[k] = arena + slot[q];
    N.set_len[k] = (uint32_t)(N.set_len[plo[q]] + N.set_len[phi[q]] - dups);
}

// ---- unions by merge path (the default) -----------------------------------------------------------------------------------------
// The merged sequence of a pair (A first among equals) is cut into tiles of 1 024 elements; a tile's share of A and of B is found
// by a binary search on its diagonal, both shares are read ONCE, consecutively, into LDS, every thread merges four elements, and an
// element of B equal to the element of A in front of it is dropped.
constexpr uint32_t MP_TILE = 1024;
__global__ void hm_ntiles_kernel(uint32_t P, const uint32_t *wlen, uint32_t *ntile)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < P) ntile[q] = (wlen[q] + MP_TILE - 1) / MP_TILE;
}
__global__ void hm_heads_kernel(uint32_t P, const uint32_t *ntile, const uint32_t *tscan, uint32_t *tmap)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < P && ntile[q]) tmap[tscan[q]] = q;
}
__device__ inline uint32_t mp_split(const uint32_t *A, uint32_t la, const uint32_t *B, uint32_t lb, uint32_t d)
{
    uint32_t lo = d > lb ? d - lb : 0u, hi = d < la ? d : la;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ void hm_partition_kernel(uint32_t NT, const uint32_t *tmap, const uint32_t *tscan, const uint32_t *plo, const uint32_t *phi, Nodes N,
                                    uint32_t *split)
{
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= NT) return;
    const uint32_t q = tmap[tile];
    const uint32_t d = (tile - tscan[q]) * MP_TILE;
    split[tile] = mp_split(N.set_p[plo[q]], N.set_len[plo[q]], N.set_p[phi[q]], N.set_len[phi[q]], d);
}
template <bool EMIT>
__global__ __launch_bounds__(256) void hm_merge_kernel(uint32_t NT, const uint32_t *tmap, const uint32_t *tscan, const uint32_t *split, const uint32_t *plo,
                                                       const uint32_t *phi, Nodes N, uint32_t *tile_count, const uint64_t *kscan, const uint64_t *slot,
                                                       uint32_t *arena)
{
    __shared__ uint32_t sA[MP_TILE], sB[MP_TILE];
    __shared__ uint32_t s_w[4];
    const uint32_t tile = blockIdx.x;
    if (tile >= NT) return;
    const uint32_t q = tmap[tile];
    const uint32_t *A = N.set_p[plo[q]], *B = N.set_p[phi[q]];
    const uint32_t la = N.set_len[plo[q]], lb = N.set_len[phi[q]];
    const uint32_t lt = tile - tscan[q], d0 = lt * MP_TILE, d1 = min(d0 + MP_TILE, la + lb);
    const uint32_t i0 = split[tile], i1 = (tile + 1 < NT && tmap[tile + 1] == q) ? split[tile + 1] : la;
    const uint32_t j0 = d0 - i0, j1 = d1 - i1;
    const uint32_t na = i1 - i0, nb = j1 - j0;
    for (uint32_t i = threadIdx.x; i < na; i += blockDim.x) sA[i] = A[i0 + i];
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) sB[i] = B[j0 + i];
    const bool have_prev = i0 > 0;
    const uint32_t prevA = have_prev ? A[i0 - 1] : 0u;
    __syncthreads();
    const uint32_t p0 = threadIdx.x * 4u, total = na + nb;
    uint32_t keep[4], val[4], cnt = 0;
    if (p0 < total) {
        uint32_t ia = mp_split(sA, na, sB, nb, p0), ib = p0 - ia;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            keep[e] = 0;
            val[e] = 0;
            if (p0 + e < total) {
                const bool fromA = ia < na && (ib >= nb || sA[ia] <= sB[ib]);
                if (fromA) {
                    val[e] = sA[ia++];
                    keep[e] = 1;
                } else {
                    const uint32_t v = sB[ib++];
                    const bool dup = ia > 0 ? sA[ia - 1] == v : (have_prev && prevA == v);
                    val[e] = v;
                    keep[e] = dup ? 0u : 1u;
                }
                cnt += keep[e];
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) keep[e] = 0, val[e] = 0;
    }
    // exclusive prefix of cnt over the block
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wbase += s_w[w];
    const uint32_t excl = wbase + incl - cnt;
    if (!EMIT) {
        if (threadIdx.x == blockDim.x - 1) tile_count[tile] = excl + cnt;
    } else {
        uint32_t *out = arena + slot[q] + (kscan[tile] - kscan[tscan[q]]) + excl;
        uint32_t k = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (keep[e]) out[k++] = val[e];
    }
}
__global__ void hm_finish_kernel(uint32_t P, uint32_t base, const uint32_t *tscan, const uint64_t *kscan, const uint64_t *slot, uint32_t *arena, Nodes N)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    const uint32_t k = base + q;
    N.set_p[k] = arena + slot[q];
    N.set_len[k] = (uint32_t)(kscan[tscan[q + 1]] - kscan[tscan[q]]);
}

// ---- candidates -------------------------------------------------------------------------------------------------------------
// the live neighbours of both halves of every merge, new ids applied: (merge, neighbour) keys, sorted and made unique
__global__ void hd_cand_flag_kernel(uint64_t ne, Edges E, Nodes N, uint32_t *flag)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const uint32_t s = E.src[e], d = E.dst[e];
    uint32_t f = 0;
    if (N.alive[s] && N.alive[d] && N.into[s] != 0 && N.into[d] != N.into[s]) f = 1;  // (d is not one of the merge's own halves)
    flag[e] = f;
}
__global__ void hd_cand_emit_kernel(uint64_t ne, Edges E, Nodes N, uint32_t base, const uint32_t *flag, const uint32_t *pos, unsigned long long *key)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne || !flag[e]) return;
    const uint32_t s = E.src[e], d = E.dst[e];
    const uint32_t l = N.into[d] ? N.into[d] : d;
    key[pos[e]] = ((unsigned long long)(N.into[s] - base) << 32) | l;
}
__global__ void hd_uniq_flag_kernel(uint32_t C, const unsigned long long *key, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    flag[i] = i == 0 || key[i] != key[i - 1];
}
__global__ void hd_tasks_kernel(uint32_t C, const unsigned long long *key, const uint32_t *flag, const uint32_t *pos, uint32_t base, uint32_t *tx,
                                uint32_t *ty)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C || !flag[i]) return;
    tx[pos[i]] = base + (uint32_t)(key[i] >> 32);
    ty[pos[i]] = (uint32_t)key[i];
}
// the round is applied: halves die, the new nodes live
__global__ void hd_apply_kernel(uint32_t P, uint32_t base, const uint32_t *plo, const uint32_t *phi, Nodes N)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    N.alive[plo[q]] = N.alive[phi[q]] = 0;
    N.into[plo[q]] = N.into[phi[q]] = 0;
    N.alive[base + q] = 1;
}
__global__ void hd_live_flag_kernel(uint64_t ne, Edges E, Nodes N, uint32_t *flag)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    flag[e] = N.alive[E.src[e]] && N.alive[E.dst[e]];
}
__global__ void hd_compact_kernel(uint64_t ne, Edges E, const uint32_t *flag, const uint32_t *pos, Edges O)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne || !flag[e]) return;
    O.src[pos[e]] = E.src[e];
    O.dst[pos[e]] = E.dst[e];
    O.sim[pos[e]] = E.sim[e];
}

struct EdgeStore {
    DevBuf<uint32_t> src, dst;
    DevBuf<float> sim;
    size_t cap = 0;
    polee_status alloc(polee_ctx *ctx, size_t n)
    {
        POLEE_TRY(src.alloc(ctx, n));
        POLEE_TRY(dst.alloc(ctx, n));
        POLEE_TRY(sim.alloc(ctx, n));
        cap = n;
        return POLEE_OK;
    }
    Edges view() { return Edges{src.p, dst.p, sim.p}; }
};

inline unsigned grid_for(uint64_t n, unsigned tb = 256) { return (unsigned)((n + tb - 1) / tb); }

// |X n Y| and the similarity of every task
polee_status run_tasks(polee_ctx *ctx, Scratch &tmp, uint32_t T, const uint32_t *tx, const uint32_t *ty, Nodes N, DevBuf<float> &sim)
{
    hipStream_t stream = ctx->stream;
    POLEE_TRY(sim.alloc(ctx, (size_t)T + 1));
    if (T == 0) return POLEE_OK;
    DevBuf<uint32_t> nblk, bscan, count;
    POLEE_TRY(nblk.alloc(ctx, (size_t)T + 1));
    POLEE_TRY(bscan.alloc(ctx, (size_t)T + 1));
    POLEE_TRY(count.alloc(ctx, (size_t)T));
    HD_HIP(hipMemsetAsync(nblk.p + T, 0, 4, stream));
    HD_HIP(hipMemsetAsync(count.p, 0, (size_t)T * 4, stream));
    hipLaunchKernelGGL(hd_task_blocks_kernel, dim3(grid_for(T)), dim3(256), 0, stream, T, tx, ty, N, nblk.p);
    POLEE_KERNEL_CHECK(ctx);
    HD_HIP(exclusive_sum(tmp, nblk.p, bscan.p, 0u, (size_t)T + 1, stream));
    uint32_t nblocks = 0;
    HD_HIP(hipMemcpyAsync(&nblocks, bscan.p + T, 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipStreamSynchronize(stream));
    if (nblocks) {  // (pieces of work: four to a block)
        DevBuf<uint32_t> pmap;
        POLEE_TRY(pmap.alloc(ctx, nblocks));
        HD_HIP(hipMemsetAsync(pmap.p, 0, (size_t)nblocks * 4, stream));
        hipLaunchKernelGGL(hd_piece_heads_kernel, dim3(grid_for(T)), dim3(256), 0, stream, T, nblk.p, bscan.p, pmap.p);
        {
            size_t bytes = 0;
            HD_HIP(rocprim::inclusive_scan(nullptr, bytes, pmap.p, pmap.p, (size_t)nblocks, rocprim::maximum<uint32_t>(), stream));
            HD_HIP(tmp.need(bytes));
            HD_HIP(rocprim::inclusive_scan(tmp.p, bytes, pmap.p, pmap.p, (size_t)nblocks, rocprim::maximum<uint32_t>(), stream));
        }
        hipLaunchKernelGGL(hd_isect_kernel, dim3((nblocks + 3) / 4), dim3(256), 0, stream, nblocks, pmap.p, tx, ty, N, bscan.p, count.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    hipLaunchKernelGGL(hd_sim_kernel, dim3(grid_for(T)), dim3(256), 0, stream, T, tx, ty, N, count.p, sim.p);
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

// the edges of a task list appended to E at `ne` (grown when needed)
polee_status append_edges(polee_ctx *ctx, Scratch &tmp, uint32_t T, const uint32_t *tx, const uint32_t *ty, const float *sim, uint32_t base, EdgeStore &E,
                          uint64_t &ne)
{
    hipStream_t stream = ctx->stream;
    if (T == 0) return POLEE_OK;
    DevBuf<uint32_t> cnt, pos;
    POLEE_TRY(cnt.alloc(ctx, (size_t)T + 1));
    POLEE_TRY(pos.alloc(ctx, (size_t)T + 1));
    HD_HIP(hipMemsetAsync(cnt.p + T, 0, 4, stream));
    hipLaunchKernelGGL(hd_edge_count_kernel, dim3(grid_for(T)), dim3(256), 0, stream, T, ty, sim, base, cnt.p);
    POLEE_KERNEL_CHECK(ctx);
    HD_HIP(exclusive_sum(tmp, cnt.p, pos.p, 0u, (size_t)T + 1, stream));
    uint32_t add = 0;
    HD_HIP(hipMemcpyAsync(&add, pos.p + T, 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipStreamSynchronize(stream));
    if (ne + add > E.cap) {
        EdgeStore G;
        POLEE_TRY(G.alloc(ctx, (size_t)((ne + add) * 3 / 2 + 1024)));
        if (ne) {
            HD_HIP(hipMemcpyAsync(G.src.p, E.src.p, ne * 4, hipMemcpyDeviceToDevice, stream));
            HD_HIP(hipMemcpyAsync(G.dst.p, E.dst.p, ne * 4, hipMemcpyDeviceToDevice, stream));
            HD_HIP(hipMemcpyAsync(G.sim.p, E.sim.p, ne * 4, hipMemcpyDeviceToDevice, stream));
            HD_HIP(hipStreamSynchronize(stream));
        }
        E.src.take(G.src);
        E.dst.take(G.dst);
        E.sim.take(G.sim);
        E.cap = G.cap;
    }
    hipLaunchKernelGGL(hd_edge_emit_kernel, dim3(grid_for(T)), dim3(256), 0, stream, T, tx, ty, sim, base, pos.p, ne, E.view());
    POLEE_KERNEL_CHECK(ctx);
    ne += add;
    return POLEE_OK;
}

}  // namespace

// X by columns (CSC, 1-based) in DEVICE memory -> the tree
static polee_status hclust_rounds_device_core(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *d_cp_p, uint64_t nnz, const uint32_t *d_rowval_p,
                                              int32_t *node_parent_idxs, int32_t *node_js)
{
    hipStream_t stream = ctx->stream;
    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(stream);
        fprintf(stderr, "[hclust/device] %-24s %.3f s\n", what, now() - t_prev);
        t_prev = now();
    };
    if (2 * (uint64_t)n + 1 >= (1ull << 31) || nnz >= (1ull << 32) - 1) return fail(ctx, POLEE_ERR_UNSUPPORTED, "hclust (device): matrix too large");
    const int K = 25;
    const uint32_t cap = 2 * (uint32_t)n + 1;
    Scratch tmp(ctx);
    struct {
        const uint64_t *p;
    } d_cp{d_cp_p};
    struct {
        const uint32_t *p;
    } d_rowval{d_rowval_p};
    DevBuf<uint32_t> err;
    POLEE_TRY(err.alloc(ctx, 1));
    HD_HIP(hipMemsetAsync(err.p, 0, 4, stream));
    // ---- leaves in the order of their median compatible read (hclust.jl:204-222; the sort is stable); columns validated
    DevBuf<uint32_t> med, idx, med_s, idxs, leaf_t;
    POLEE_TRY(med.alloc(ctx, (size_t)n));
    POLEE_TRY(idx.alloc(ctx, (size_t)n));
    POLEE_TRY(med_s.alloc(ctx, (size_t)n));
    POLEE_TRY(idxs.alloc(ctx, (size_t)n));
    POLEE_TRY(leaf_t.alloc(ctx, (size_t)n));
    hipLaunchKernelGGL(hd_median_kernel, dim3(grid_for((uint64_t)n)), dim3(256), 0, stream, n, m, d_cp.p, d_rowval.p, med.p, idx.p, err.p);
    POLEE_KERNEL_CHECK(ctx);
    if (nnz) {
        DevBuf<uint32_t> colstart;
        POLEE_TRY(colstart.alloc(ctx, (size_t)nnz));
        HD_HIP(hipMemsetAsync(colstart.p, 0, (size_t)nnz * 4, stream));
        hipLaunchKernelGGL(hd_colstart_kernel, dim3(grid_for((uint64_t)n)), dim3(256), 0, stream, n, d_cp.p, nnz, colstart.p);
        POLEE_KERNEL_CHECK(ctx);
        hipLaunchKernelGGL(hd_validate_kernel, dim3(grid_for(nnz)), dim3(256), 0, stream, nnz, d_rowval.p, colstart.p, err.p);
        POLEE_KERNEL_CHECK(ctx);
        HD_HIP(hipStreamSynchronize(stream));
    }
    uint32_t h_err = 0;
    HD_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipStreamSynchronize(stream));
    if (h_err == 1) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: row indexes of a column are not ascending");
    if (h_err == 2) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: row index out of range");
    HD_HIP(sort_pairs(tmp, med.p, med_s.p, idx.p, idxs.p, (size_t)n, false, stream));
    // ---- node arrays
    DevBuf<const uint32_t *> set_p;
    DevBuf<uint32_t> set_len, into, best;
    DevBuf<uint8_t> alive;
    DevBuf<unsigned long long> bestkey, besttie;
    DevBuf<int32_t> left, right;
    POLEE_TRY(set_p.alloc(ctx, cap));
    POLEE_TRY(set_len.alloc(ctx, cap));
    POLEE_TRY(into.alloc(ctx, cap));
    POLEE_TRY(best.alloc(ctx, cap));
    POLEE_TRY(alive.alloc(ctx, cap));
    POLEE_TRY(bestkey.alloc(ctx, cap));
    POLEE_TRY(besttie.alloc(ctx, cap));
    POLEE_TRY(left.alloc(ctx, cap));
    POLEE_TRY(right.alloc(ctx, cap));
    HD_HIP(hipMemsetAsync(set_p.p, 0, (size_t)cap * sizeof(void *), stream));
    HD_HIP(hipMemsetAsync(set_len.p, 0, (size_t)cap * 4, stream));
    HD_HIP(hipMemsetAsync(into.p, 0, (size_t)cap * 4, stream));
    HD_HIP(hipMemsetAsync(alive.p, 0, (size_t)cap, stream));
    HD_HIP(hipMemsetAsync(left.p, 0xff, (size_t)cap * 4, stream));
    HD_HIP(hipMemsetAsync(right.p, 0xff, (size_t)cap * 4, stream));
    Nodes N{set_p.p, set_len.p, alive.p, into.p, bestkey.p, besttie.p, best.p, left.p, right.p};
    hipLaunchKernelGGL(hd_leaves_kernel, dim3(grid_for((uint64_t)n)), dim3(256), 0, stream, n, d_cp.p, d_rowval.p, idxs.p, N, leaf_t.p);
    POLEE_KERNEL_CHECK(ctx);
    // ---- similarities of every leaf to its K successors, the initial edges
    EdgeStore E;
    uint64_t ne = 0;
    {
        const uint64_t T64 = (uint64_t)n * K;
        if (T64 >= (1ull << 32)) return fail(ctx, POLEE_ERR_UNSUPPORTED, "hclust (device): matrix too large");
        const uint32_t T = (uint32_t)T64;
        DevBuf<uint32_t> tx, ty;
        DevBuf<float> sim;
        POLEE_TRY(tx.alloc(ctx, T));
        POLEE_TRY(ty.alloc(ctx, T));
        hipLaunchKernelGGL(hd_leaf_tasks_kernel, dim3(grid_for(T)), dim3(256), 0, stream, n, K, tx.p, ty.p);
        POLEE_KERNEL_CHECK(ctx);
        POLEE_TRY(run_tasks(ctx, tmp, T, tx.p, ty.p, N, sim));
        lap("leaves + similarities");
        POLEE_TRY(E.alloc(ctx, (size_t)2 * T + 4 * (size_t)n + 1024));
        POLEE_TRY(append_edges(ctx, tmp, T, tx.p, ty.p, sim.p, 0xffffffffu, E, ne));  // (both directions: every leaf is "old")
        lap("initial edges");
    }
    // ---- rounds
    // one arena per round: a merged node's reads are a slice of its round's arena, which goes back (to the kept device buffers: the
    // next round's arena is usually one of them) when the last of its nodes has been merged away
    std::deque<DevBuf<uint32_t>> arenas;
    std::vector<uint32_t> arena_first_id, arena_live;  // nodes [arena_first_id[r], arena_first_id[r + 1]) live in arena r
    std::vector<uint32_t> h_plo, h_phi;
    uint32_t next_id = (uint32_t)n + 1;
    size_t rounds = 0, n_eval = 0;
    EdgeStore E2;
    DevBuf<uint32_t> flag, pos, plo, phi, wlen, blen;
    DevBuf<unsigned long long> pkey, ptie, pkey2, ptie2;
    POLEE_TRY(flag.alloc(ctx, (size_t)cap + 1));
    POLEE_TRY(pos.alloc(ctx, (size_t)cap + 1));
    while (ne > 0) {
        ++rounds;
        HD_HIP(hipMemsetAsync(bestkey.p, 0, (size_t)cap * 8, stream));
        HD_HIP(hipMemsetAsync(besttie.p, 0xff, (size_t)cap * 8, stream));
        hipLaunchKernelGGL(hd_best1_kernel, dim3(grid_for(ne)), dim3(256), 0, stream, ne, E.view(), N);
        hipLaunchKernelGGL(hd_best2_kernel, dim3(grid_for(ne)), dim3(256), 0, stream, ne, E.view(), N);
        hipLaunchKernelGGL(hd_best3_kernel, dim3(grid_for(cap)), dim3(256), 0, stream, cap, N);
        HD_HIP(hipMemsetAsync(flag.p + cap, 0, 4, stream));
        hipLaunchKernelGGL(hd_mutual_kernel, dim3(grid_for(cap)), dim3(256), 0, stream, cap, N, flag.p);
        POLEE_KERNEL_CHECK(ctx);
        HD_HIP(exclusive_sum(tmp, flag.p, pos.p, 0u, (size_t)cap + 1, stream));
        uint32_t P = 0;
        HD_HIP(hipMemcpyAsync(&P, pos.p + cap, 4, hipMemcpyDeviceToHost, stream));
        HD_HIP(hipStreamSynchronize(stream));
        if (P == 0) break;  // (cannot happen while an edge is left: the best edge overall is mutual)
        if ((uint64_t)next_id + P > cap) return fail(ctx, POLEE_ERR_HIP, "hclust (device): internal error: node count");
        POLEE_TRY(pkey.alloc(ctx, P));
        POLEE_TRY(ptie.alloc(ctx, P));
        POLEE_TRY(pkey2.alloc(ctx, P));
        POLEE_TRY(ptie2.alloc(ctx, P));
        hipLaunchKernelGGL(hd_pairs_kernel, dim3(grid_for(cap)), dim3(256), 0, stream, cap, N, flag.p, pos.p, pkey.p, ptie.p);
        POLEE_KERNEL_CHECK(ctx);
        // descending by key; equal keys by ascending (lo, hi): a stable sort on the key after a sort on (lo, hi)
        HD_HIP(sort_pairs(tmp, ptie.p, ptie2.p, pkey.p, pkey2.p, (size_t)P, false, stream));
        HD_HIP(sort_pairs(tmp, pkey2.p, pkey.p, ptie2.p, ptie.p, (size_t)P, true, stream));
        const uint32_t base = next_id;
        next_id += P;
        POLEE_TRY(plo.alloc(ctx, (size_t)P + 1));
        POLEE_TRY(phi.alloc(ctx, (size_t)P + 1));
        POLEE_TRY(wlen.alloc(ctx, (size_t)P + 1));
        POLEE_TRY(blen.alloc(ctx, (size_t)P + 1));
        hipLaunchKernelGGL(hd_ids_kernel, dim3(grid_for(P)), dim3(256), 0, stream, P, ptie.p, base, N, plo.p, phi.p, wlen.p, blen.p);
        POLEE_KERNEL_CHECK(ctx);
        // ---- unions into this round's arena
        DevBuf<uint64_t> slot, bscan, ascan, dscan;
        POLEE_TRY(slot.alloc(ctx, (size_t)P + 1));
        POLEE_TRY(bscan.alloc(ctx, (size_t)P + 1));
        POLEE_TRY(ascan.alloc(ctx, (size_t)P + 1));
        HD_HIP(hipMemsetAsync(wlen.p + P, 0, 4, stream));
        HD_HIP(hipMemsetAsync(blen.p + P, 0, 4, stream));
        HD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(wlen.p, ToU64()), slot.p, (uint64_t)0, (size_t)P + 1, stream));
        HD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(blen.p, ToU64()), bscan.p, (uint64_t)0, (size_t)P + 1, stream));
        uint64_t tot_w = 0, tot_b = 0;
        HD_HIP(hipMemcpyAsync(&tot_w, slot.p + P, 8, hipMemcpyDeviceToHost, stream));
        HD_HIP(hipMemcpyAsync(&tot_b, bscan.p + P, 8, hipMemcpyDeviceToHost, stream));
        HD_HIP(hipStreamSynchronize(stream));
        const uint64_t tot_a = tot_w - tot_b;
        // ascan = slot - bscan (the scan of A's lengths)
        hipLaunchKernelGGL(hd_diff_u64_kernel, dim3(grid_for((uint64_t)P + 1)), dim3(256), 0, stream, P + 1, slot.p, bscan.p, ascan.p);
        POLEE_KERNEL_CHECK(ctx);
        arenas.emplace_back();
        arena_first_id.push_back(base);
        arena_live.push_back(P);
        DevBuf<uint32_t> &arena = arenas.back();
        POLEE_TRY(arena.alloc(ctx, (size_t)tot_w + 1));
        // (default; POLEE_HCLUST_MERGE_PATH=0: a binary search per element instead -- 19 ms of kernels per tree at C2 against 3.9 ms)
        static const bool merge_path = getenv("POLEE_HCLUST_MERGE_PATH") == nullptr || atoi(getenv("POLEE_HCLUST_MERGE_PATH")) != 0;
        if (merge_path) {
            DevBuf<uint32_t> ntile, tscan, tmap, split, tcount;
            DevBuf<uint64_t> kscan;
            POLEE_TRY(ntile.alloc(ctx, (size_t)P + 1));
            POLEE_TRY(tscan.alloc(ctx, (size_t)P + 1));
            HD_HIP(hipMemsetAsync(ntile.p + P, 0, 4, stream));
            hipLaunchKernelGGL(hm_ntiles_kernel, dim3(grid_for(P)), dim3(256), 0, stream, P, wlen.p, ntile.p);
            POLEE_KERNEL_CHECK(ctx);
            HD_HIP(exclusive_sum(tmp, ntile.p, tscan.p, 0u, (size_t)P + 1, stream));
            uint32_t NT = 0;
            HD_HIP(hipMemcpyAsync(&NT, tscan.p + P, 4, hipMemcpyDeviceToHost, stream));
            HD_HIP(hipStreamSynchronize(stream));
            POLEE_TRY(tmap.alloc(ctx, (size_t)NT + 1));
            POLEE_TRY(split.alloc(ctx, (size_t)NT + 1));
            POLEE_TRY(tcount.alloc(ctx, (size_t)NT + 1));
            POLEE_TRY(kscan.alloc(ctx, (size_t)NT + 1));
            HD_HIP(hipMemsetAsync(tmap.p, 0, ((size_t)NT + 1) * 4, stream));
            HD_HIP(hipMemsetAsync(tcount.p + NT, 0, 4, stream));
            hipLaunchKernelGGL(hm_heads_kernel, dim3(grid_for(P)), dim3(256), 0, stream, P, ntile.p, tscan.p, tmap.p);
            {
                size_t bytes = 0;
                HD_HIP(rocprim::inclusive_scan(nullptr, bytes, tmap.p, tmap.p, (size_t)NT, rocprim::maximum<uint32_t>(), stream));
                HD_HIP(tmp.need(bytes));
                HD_HIP(rocprim::inclusive_scan(tmp.p, bytes, tmap.p, tmap.p, (size_t)NT, rocprim::maximum<uint32_t>(), stream));
            }
            hipLaunchKernelGGL(hm_partition_kernel, dim3(grid_for(NT)), dim3(256), 0, stream, NT, tmap.p, tscan.p, plo.p, phi.p, N, split.p);
            hipLaunchKernelGGL((hm_merge_kernel<false>), dim3(NT), dim3(256), 0, stream, NT, tmap.p, tscan.p, split.p, plo.p, phi.p, N, tcount.p,
                               (const uint64_t *)nullptr, slot.p, arena.p);
            POLEE_KERNEL_CHECK(ctx);
            HD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(tcount.p, ToU64()), kscan.p, (uint64_t)0, (size_t)NT + 1, stream));
            hipLaunchKernelGGL((hm_merge_kernel<true>), dim3(NT), dim3(256), 0, stream, NT, tmap.p, tscan.p, split.p, plo.p, phi.p, N, tcount.p, kscan.p,
                               slot.p, arena.p);
            hipLaunchKernelGGL(hm_finish_kernel, dim3(grid_for(P)), dim3(256), 0, stream, P, base, tscan.p, kscan.p, slot.p, arena.p, N);
            POLEE_KERNEL_CHECK(ctx);
            HD_HIP(hipStreamSynchronize(stream));
        } else {
            DevBuf<uint32_t> isdup, lba;
            POLEE_TRY(isdup.alloc(ctx, (size_t)tot_b + 1));
            POLEE_TRY(lba.alloc(ctx, (size_t)tot_b + 1));
            POLEE_TRY(dscan.alloc(ctx, (size_t)tot_b + 1));
            HD_HIP(hipMemsetAsync(isdup.p + tot_b, 0, 4, stream));
            if (tot_b) {
                hipLaunchKernelGGL(hd_union_b_kernel, dim3(grid_for(tot_b)), dim3(256), 0, stream, tot_b, P, bscan.p, plo.p, phi.p, N, isdup.p, lba.p);
                POLEE_KERNEL_CHECK(ctx);
            }
            HD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(isdup.p, ToU64()), dscan.p, (uint64_t)0, (size_t)tot_b + 1, stream));
            if (tot_b) {
                hipLaunchKernelGGL(hd_union_place_b_kernel, dim3(grid_for(tot_b)), dim3(256), 0, stream, tot_b, P, bscan.p, phi.p, N, isdup.p, dscan.p, lba.p,
                                   slot.p, arena.p);
                POLEE_KERNEL_CHECK(ctx);
            }
            if (tot_a) {
                hipLaunchKernelGGL(hd_union_place_a_kernel, dim3(grid_for(tot_a)), dim3(256), 0, stream, tot_a, P, ascan.p, bscan.p, plo.p, phi.p, N, dscan.p,
                                   slot.p, arena.p);
                POLEE_KERNEL_CHECK(ctx);
            }
            hipLaunchKernelGGL(hd_union_finish_kernel, dim3(grid_for(P)), dim3(256), 0, stream, P, base, plo.p, phi.p, bscan.p, dscan.p, slot.p, arena.p, N);
            POLEE_KERNEL_CHECK(ctx);
        }
        // ---- candidates: the live neighbours of both halves
        DevBuf<uint32_t> eflag, epos;
        POLEE_TRY(eflag.alloc(ctx, (size_t)ne + 1));
        POLEE_TRY(epos.alloc(ctx, (size_t)ne + 1));
        HD_HIP(hipMemsetAsync(eflag.p + ne, 0, 4, stream));
        hipLaunchKernelGGL(hd_cand_flag_kernel, dim3(grid_for(ne)), dim3(256), 0, stream, ne, E.view(), N, eflag.p);
        POLEE_KERNEL_CHECK(ctx);
        HD_HIP(exclusive_sum(tmp, eflag.p, epos.p, 0u, (size_t)ne + 1, stream));
        uint32_t C = 0;
        HD_HIP(hipMemcpyAsync(&C, epos.p + ne, 4, hipMemcpyDeviceToHost, stream));
        HD_HIP(hipStreamSynchronize(stream));
        uint32_t T = 0;
        DevBuf<uint32_t> tx, ty;
        DevBuf<float> sim;
        if (C) {
            DevBuf<unsigned long long> ckey, ckey_s;
            DevBuf<uint32_t> dummy, dummy_s, uflag, upos;
            POLEE_TRY(ckey.alloc(ctx, C));
            POLEE_TRY(ckey_s.alloc(ctx, C));
            hipLaunchKernelGGL(hd_cand_emit_kernel, dim3(grid_for(ne)), dim3(256), 0, stream, ne, E.view(), N, base, eflag.p, epos.p, ckey.p);
            POLEE_KERNEL_CHECK(ctx);
            {
                size_t bytes = 0;
                HD_HIP(rocprim::radix_sort_keys(nullptr, bytes, ckey.p, ckey_s.p, (size_t)C, 0, 64, stream));
                HD_HIP(tmp.need(bytes));
                HD_HIP(rocprim::radix_sort_keys(tmp.p, bytes, ckey.p, ckey_s.p, (size_t)C, 0, 64, stream));
            }
            POLEE_TRY(uflag.alloc(ctx, (size_t)C + 1));
            POLEE_TRY(upos.alloc(ctx, (size_t)C + 1));
            HD_HIP(hipMemsetAsync(uflag.p + C, 0, 4, stream));
            hipLaunchKernelGGL(hd_uniq_flag_kernel, dim3(grid_for(C)), dim3(256), 0, stream, C, ckey_s.p, uflag.p);
            POLEE_KERNEL_CHECK(ctx);
            HD_HIP(exclusive_sum(tmp, uflag.p, upos.p, 0u, (size_t)C + 1, stream));
            HD_HIP(hipMemcpyAsync(&T, upos.p + C, 4, hipMemcpyDeviceToHost, stream));
            HD_HIP(hipStreamSynchronize(stream));
            POLEE_TRY(tx.alloc(ctx, (size_t)T + 1));
            POLEE_TRY(ty.alloc(ctx, (size_t)T + 1));
            hipLaunchKernelGGL(hd_tasks_kernel, dim3(grid_for(C)), dim3(256), 0, stream, C, ckey_s.p, uflag.p, upos.p, base, tx.p, ty.p);
            POLEE_KERNEL_CHECK(ctx);
            // ---- similarities of the new nodes to their candidates
            POLEE_TRY(run_tasks(ctx, tmp, T, tx.p, ty.p, N, sim));
            n_eval += T;
        }
        // ---- the halves retire, the new nodes live; dead edges go, the new ones are appended
        hipLaunchKernelGGL(hd_apply_kernel, dim3(grid_for(P)), dim3(256), 0, stream, P, base, plo.p, phi.p, N);
        POLEE_KERNEL_CHECK(ctx);
        hipLaunchKernelGGL(hd_live_flag_kernel, dim3(grid_for(ne)), dim3(256), 0, stream, ne, E.view(), N, eflag.p);
        POLEE_KERNEL_CHECK(ctx);
        HD_HIP(exclusive_sum(tmp, eflag.p, epos.p, 0u, (size_t)ne + 1, stream));
        uint32_t nlive = 0;
        HD_HIP(hipMemcpyAsync(&nlive, epos.p + ne, 4, hipMemcpyDeviceToHost, stream));
        HD_HIP(hipStreamSynchronize(stream));
        if (E2.cap < E.cap) POLEE_TRY(E2.alloc(ctx, E.cap));
        hipLaunchKernelGGL(hd_compact_kernel, dim3(grid_for(ne)), dim3(256), 0, stream, ne, E.view(), eflag.p, epos.p, E2.view());
        POLEE_KERNEL_CHECK(ctx);
        {
            EdgeStore H;  // E <-> E2 (ownership moves with the blocks)
            H.src.take(E.src); H.dst.take(E.dst); H.sim.take(E.sim); H.cap = E.cap;
            E.src.take(E2.src); E.dst.take(E2.dst); E.sim.take(E2.sim); E.cap = E2.cap;
            E2.src.take(H.src); E2.dst.take(H.dst); E2.sim.take(H.sim); E2.cap = H.cap;
        }
        ne = nlive;
        if (T) POLEE_TRY(append_edges(ctx, tmp, T, tx.p, ty.p, sim.p, base, E, ne));
        h_plo.resize(P);
        h_phi.resize(P);
        HD_HIP(hipMemcpyAsync(h_plo.data(), plo.p, (size_t)P * 4, hipMemcpyDeviceToHost, stream));
        HD_HIP(hipMemcpyAsync(h_phi.data(), phi.p, (size_t)P * 4, hipMemcpyDeviceToHost, stream));
        HD_HIP(hipStreamSynchronize(stream));
        for (int half = 0; half < 2; ++half)
            for (uint32_t x : half ? h_phi : h_plo) {
                if (x <= (uint32_t)n) continue;  // (a leaf: its reads are a column of X)
                const size_t r = (size_t)(std::upper_bound(arena_first_id.begin(), arena_first_id.end(), x) - arena_first_id.begin()) - 1;
                if (--arena_live[r] == 0) arenas[r].release();
            }
    }
    if (timing) fprintf(stderr, "[hclust/device]   %zu rounds, %zu similarity evaluations\n", rounds, n_eval);
    lap("joining in rounds");
    // ---- what is left is joined smallest first, then the nodes are ordered: on the host (2 n small records)
    std::vector<int32_t> h_left(next_id), h_right(next_id);
    std::vector<uint8_t> h_alive(next_id);
    std::vector<uint32_t> h_len(next_id), h_leaf_t((size_t)n);
    HD_HIP(hipMemcpyAsync(h_left.data(), left.p, (size_t)next_id * 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipMemcpyAsync(h_right.data(), right.p, (size_t)next_id * 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipMemcpyAsync(h_alive.data(), alive.p, (size_t)next_id, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipMemcpyAsync(h_len.data(), set_len.p, (size_t)next_id * 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipMemcpyAsync(h_leaf_t.data(), leaf_t.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipStreamSynchronize(stream));
    const std::string e = hclust_finish_from_arrays(n, next_id, h_left.data(), h_right.data(), h_alive.data(), h_len.data(), h_leaf_t.data(), node_parent_idxs,
                                                    node_js);
    if (!e.empty()) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: %s", e.c_str());
    lap("remaining joins + order");
    return POLEE_OK;
}

polee_status hclust_rounds_device(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                  int32_t *node_parent_idxs, int32_t *node_js)
{
    if (n < 1 || m < 0 || !colptr || !node_parent_idxs || !node_js) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: bad argument");
    if (colptr_bytes != 4 && colptr_bytes != 8) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: colptr_bytes must be 4 or 8");
    POLEE_TRY(use_device(ctx));
    std::vector<uint64_t> cp((size_t)n + 1);
    for (int64_t j = 0; j <= n; ++j)
        cp[(size_t)j] = colptr_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(colptr)[j] : reinterpret_cast<const uint64_t *>(colptr)[j];
    if (cp[0] != 1) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: colptr[0] must be 1 (1-based)");
    for (int64_t j = 0; j < n; ++j)
        if (cp[(size_t)j + 1] < cp[(size_t)j]) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: colptr is not monotone");
    const uint64_t nnz = cp[(size_t)n] - 1;
    if (nnz >= (1ull << 32) - 1) return fail(ctx, POLEE_ERR_UNSUPPORTED, "hclust (device): matrix too large");
    DevBuf<uint64_t> d_cp;
    DevBuf<uint32_t> d_rowval;
    POLEE_TRY(d_cp.upload(ctx, cp.data(), cp.size()));
    POLEE_TRY(d_rowval.upload(ctx, rowval, (size_t)nnz));
    return hclust_rounds_device_core(ctx, m, n, d_cp.p, nnz, d_rowval.p, node_parent_idxs, node_js);
}

namespace {
// Xt (the rows of X, 1-based, device memory) -> X by columns: entries counted per transcript, a STABLE sort by transcript of
// (transcript, fragment) pairs -- the fragments of a column come out ascending, as the clustering needs them
__global__ void hx_count_kernel(uint64_t nnz, int64_t n, const uint32_t *trowval, uint32_t *counts, uint32_t *key, uint32_t *err)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const uint32_t t = trowval[k];
    if (t < 1 || (int64_t)t > n) {
        atomicMax(err, 1u);
        key[k] = 0;
        return;
    }
    key[k] = t - 1u;
    atomicAdd(&counts[t - 1u], 1u);
}
__global__ void hx_rowids_kernel(int64_t m, const uint64_t *tcolptr, uint32_t *rowid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    for (uint64_t k = tcolptr[i] - 1; k < tcolptr[i + 1] - 1; ++k) rowid[k] = (uint32_t)i + 1u;
}
__global__ void hx_plus1_kernel(int64_t n1, uint64_t *cp)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n1) cp[j] += 1u;
}
}  // namespace

polee_status hclust_rounds_device_from_xt(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *d_tcolptr, const uint32_t *d_trowval,
                                          int32_t *node_parent_idxs, int32_t *node_js)
{
    if (n < 1 || m < 0 || !d_tcolptr || !node_parent_idxs || !node_js) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: bad argument");
    POLEE_TRY(use_device(ctx));
    hipStream_t stream = ctx->stream;
    uint64_t last = 0;
    HD_HIP(hipMemcpyAsync(&last, d_tcolptr + m, 8, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipStreamSynchronize(stream));
    const uint64_t nnz = last - 1;
    if (nnz >= (1ull << 32) - 1 || m >= ((int64_t)1 << 32) - 1) return fail(ctx, POLEE_ERR_UNSUPPORTED, "hclust (device): matrix too large");
    Scratch tmp(ctx);
    DevBuf<uint64_t> d_cp;
    DevBuf<uint32_t> counts, key, key_s, rowid, rowval, err;
    POLEE_TRY(d_cp.alloc(ctx, (size_t)n + 1));
    POLEE_TRY(counts.alloc(ctx, (size_t)n + 1));
    POLEE_TRY(key.alloc(ctx, (size_t)nnz + 1));
    POLEE_TRY(key_s.alloc(ctx, (size_t)nnz + 1));
    POLEE_TRY(rowid.alloc(ctx, (size_t)nnz + 1));
    POLEE_TRY(rowval.alloc(ctx, (size_t)nnz + 1));
    POLEE_TRY(err.alloc(ctx, 1));
    HD_HIP(hipMemsetAsync(counts.p, 0, ((size_t)n + 1) * 4, stream));
    HD_HIP(hipMemsetAsync(err.p, 0, 4, stream));
    if (nnz) {
        hipLaunchKernelGGL(hx_count_kernel, dim3(grid_for(nnz)), dim3(256), 0, stream, nnz, n, d_trowval, counts.p, key.p, err.p);
        hipLaunchKernelGGL(hx_rowids_kernel, dim3(grid_for((uint64_t)m)), dim3(256), 0, stream, m, d_tcolptr, rowid.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    HD_HIP(exclusive_sum(tmp, rocprim::make_transform_iterator(counts.p, ToU64()), d_cp.p, (uint64_t)0, (size_t)n + 1, stream));
    hipLaunchKernelGGL(hx_plus1_kernel, dim3(grid_for((uint64_t)n + 1)), dim3(256), 0, stream, n + 1, d_cp.p);
    POLEE_KERNEL_CHECK(ctx);
    uint32_t h_err = 0;
    HD_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, stream));
    HD_HIP(hipStreamSynchronize(stream));
    if (h_err) return fail(ctx, POLEE_ERR_BAD_ARG, "hclust: transcript index out of range");
    if (nnz) {
        unsigned bits = 1;
        while (bits < 32 && ((uint64_t)1 << bits) < (uint64_t)n) ++bits;
        size_t bytes = 0;
        HD_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key_s.p, rowid.p, rowval.p, (size_t)nnz, 0, bits, stream));
        HD_HIP(tmp.need(bytes));
        HD_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key_s.p, rowid.p, rowval.p, (size_t)nnz, 0, bits, stream));
    }
    key.release();
    key_s.release();
    rowid.release();
    return hclust_rounds_device_core(ctx, m, n, d_cp.p, nnz, rowval.p, node_parent_idxs, node_js);
}

}  // namespace polee

extern "C" polee_status polee_hclust_parallel_device(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                                     int32_t *node_parent_idxs, int32_t *node_js)
{
    return polee::guarded(ctx, "polee_hclust_parallel_device",
                          [&] { return polee::hclust_rounds_device(ctx, m, n, colptr, colptr_bytes, rowval, node_parent_idxs, node_js); });
}

extern "C" polee_status polee_hclust_parallel_device_from_devx(polee_ctx *ctx, const polee_devx *dx, int32_t *node_parent_idxs, int32_t *node_js)
{
    return polee::guarded(ctx, "polee_hclust_parallel_device_from_devx", [&]() -> polee_status {
        if (!ctx || !dx || !node_parent_idxs || !node_js) return polee::fail(ctx, POLEE_ERR_BAD_ARG, "polee_hclust_parallel_device_from_devx: null argument");
        if (dx->ctx->device != ctx->device) return polee::fail(ctx, POLEE_ERR_BAD_ARG, "polee_hclust_parallel_device_from_devx: X lives on another device");
        POLEE_TRY(polee::use_device(ctx));
        return polee::hclust_rounds_device_core(ctx, dx->m, dx->n, dx->cp.p, dx->nnz, dx->rowval.p, node_parent_idxs, node_js);
    });
}

extern "C" polee_status polee_hclust_parallel_device_from_xbuild(polee_ctx *ctx, const polee_xbuild *xb, int32_t *node_parent_idxs, int32_t *node_js)
{
    return polee::guarded(ctx, "polee_hclust_parallel_device_from_xbuild", [&]() -> polee_status {
        polee_ctx *xctx = nullptr;
        int64_t m = 0, n = 0;
        const uint64_t *tcolptr = nullptr;
        const uint32_t *trowval = nullptr;
        const float *tnzval = nullptr;
        POLEE_TRY(polee::xbuild_device_view(xb, &xctx, &m, &n, &tcolptr, &trowval, &tnzval));
        if (!ctx || xctx->device != ctx->device) return polee::fail(ctx, POLEE_ERR_BAD_ARG, "polee_hclust_parallel_device_from_xbuild: the xbuild result lives on another device");
        POLEE_HIP_TRY(ctx, hipStreamSynchronize(xctx->stream));
        return polee::hclust_rounds_device_from_xt(ctx, m, n, tcolptr, trowval, node_parent_idxs, node_js);
    });
}
