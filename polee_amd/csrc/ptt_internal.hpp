// Polya tree transform: host plan, device view and the scan functors shared by the
// standalone tree API (ptt.hip), the VI loop (vi.hip) and the approximation density
// (approx.hip).
//
// Layout in HBM (per tree; T trees are concatenated, all with the same n):
//   tour_code  u32 [3n-2]  Euler tour of the tree, DFS pre-order, right child first
//                          (the order the reference serialises, src/hclust.jl:361-389).
//                          bits 0-1 type (0 ENTER internal, 1 EXIT internal, 2 LEAF),
//                          bit 2 "root" (no incoming edge), bit 3 side (1 = left child,
//                          i.e. the edge multiplies by y; src/ptt.jl:147-148),
//                          bits 4.. k of the PARENT internal node (index into ys)
//   tour_tgt   i32 [3n-2]  LEAF: leaf position in DFS-leaf order; ENTER: the node's own k
//   leaf_tid   i32 [n]     leaf position -> 0-based transcript id
//   lo,mid,hi1 i32 [n-1]   per internal node k: its right subtree covers leaf positions
//                          [lo, mid), its left subtree [mid, hi1)  (right-first order)
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace polee {

constexpr uint32_t TOUR_ENTER = 0, TOUR_EXIT = 1, TOUR_LEAF = 2;

struct PttPlan {
    int32_t n = 0, N = 0;
    int64_t TL = 0;  // tour length 3n-2
    std::vector<int32_t> left, right, leaf;  // 0-based node arrays, -1 none (node order as given)
    std::vector<int32_t> node_k;             // node -> internal ordinal, -1 for leaves
    std::vector<uint32_t> tour_code;
    std::vector<int32_t> tour_tgt;
    std::vector<int32_t> leaf_tid, tid_pos;
    std::vector<int32_t> lo, mid, hi1;
    int32_t max_depth = 0;
};

// Builds the plan from 0-based child arrays; returns an error message or "".
std::string build_ptt_plan(const int32_t *left, const int32_t *right, const int32_t *leaf, int32_t N,
                           PttPlan &plan);
// src/ptt.jl:89-116 + 293-309: parent/js (1-based) -> 0-based child arrays.
std::string children_from_parents(const int32_t *node_parent_idxs, const int32_t *node_js, int32_t N,
                                  std::vector<int32_t> &left, std::vector<int32_t> &right,
                                  std::vector<int32_t> &leaf);

struct PttView {
    int32_t n, T;
    int64_t TL;
    const uint32_t *tour_code;
    const int32_t *tour_tgt;
    const int32_t *leaf_tid;
    const int32_t *lo, *mid, *hi1;
    __device__ inline int tree(int row) const { return T == 1 ? 0 : row; }
};

}  // namespace polee

struct polee_ptt {
    polee_ctx *ctx = nullptr;
    int refs = 1;
    int32_t n = 0, N = 0, T = 1;
    int64_t TL = 0;
    std::vector<polee::PttPlan> plans;  // host copies (debug / tests)
    polee::DevBuf<uint32_t> d_tour_code;
    polee::DevBuf<int32_t> d_tour_tgt, d_leaf_tid, d_lo, d_mid, d_hi1;
    // scratch, grown on demand to the largest batch seen
    int32_t cap_rows = 0;
    polee::DevBuf<polee::dd> d_chunk;  // scan chunk totals
    polee::DevBuf<double> d_ys;        // [B][n-1]
    polee::DevBuf<double> d_uleaf;     // [B][n]   leaf u in leaf order (t.us of the reference)
    polee::DevBuf<double> d_logu;      // [B][n-1] log u of internal nodes
    polee::DevBuf<polee::dd> d_C;      // [B][n+1] double-double prefix over leaf order
    polee::DevBuf<double> d_part;      // [B][nchunks][2] per-chunk partial sums
    polee::DevBuf<double> d_row;       // [B][4] per-row reductions
    polee::DevBuf<double> d_f64a, d_f64b;  // staging for the host-pointer API
    polee::DevBuf<float> d_f32a, d_f32b;

    polee::PttView view() const
    {
        return polee::PttView{n, T, TL, d_tour_code.p, d_tour_tgt.p, d_leaf_tid.p, d_lo.p, d_mid.p, d_hi1.p};
    }
    polee_status reserve(int32_t rows);
};

namespace polee {

// ---- device functors ------------------------------------------------------------------

// log of the edge factor of a tour entry: log y (left) or log(1-y) (right); 0 for the root.
// logs (optional): precomputed [2][B][n-1] = {log y, log1p(-y)}; every internal node's logs are needed
// by four tour entries in each scan phase, so callers on the hot path compute them once.
struct EdgeLogs {
    const double *ys;     // [B][n-1]
    const double *ly;     // [B][n-1] log y, or null
    const double *l1y;    // [B][n-1] log1p(-y), or null
    int64_t nm1;
    __device__ inline double operator()(uint32_t code, int row) const
    {
        if (code & 4u) return 0.0;
        const int64_t o = (int64_t)row * nm1 + (code >> 4);
        if (ly) return (code & 8u) ? ly[o] : l1y[o];
        const double y = ys[o];
        return (code & 8u) ? log(y) : log1p(-y);
    }
};

// Forward (transform!, src/ptt.jl:125-160; HSB op hsb_ops.cpp:87-109): Euler-tour scan
// of signed edge logs.  log u(node) = inclusive prefix at its ENTER; leaves add their
// own edge.
struct FwdLoad {
    PttView v;
    EdgeLogs el;
    __device__ double operator()(int row, int64_t e) const
    {
        const uint32_t code = v.tour_code[(int64_t)v.tree(row) * v.TL + e];
        const uint32_t type = code & 3u;
        if (type == TOUR_LEAF) return 0.0;
        const double lf = el(code, row);
        return type == TOUR_ENTER ? lf : -lf;
    }
};

struct FwdEmit {
    PttView v;
    EdgeLogs el;
    double *uleaf;         // [B][n] leaf order, or null
    double *logu;          // [B][n-1] by k, or null
    float *xs;             // transcript order, element (row, tid) at xs[row*xs_rs + tid*xs_es]; or null
    int64_t xs_rs, xs_es;
    double leaf_floor;     // 1e-16 (ptt.jl:139) or 0 (HSB op)
    float clamp_lo, clamp_hi;  // applied after the floor when clamp_lo > 0 (likelihood-approximation.jl:526)
    const float *efflens;  // optional: partial sum 0 accumulates x/efflen (likelihood.jl:97-100)
    int64_t efflens_rs;    // row stride of efflens (0: one vector shared by all rows)
    // returns the element's contribution to the two per-chunk partial sums
    __device__ void operator()(int row, int64_t e, double /*excl*/, double incl, double &p0, double &p1) const
    {
        const int64_t tb = (int64_t)v.tree(row) * v.TL;
        const uint32_t code = v.tour_code[tb + e];
        const uint32_t type = code & 3u;
        p0 = 0.0;
        p1 = 0.0;
        if (type == TOUR_LEAF) {
            const double lu = incl + el(code, row);
            const double u = exp(lu);
            const int pos = v.tour_tgt[tb + e];
            if (uleaf) uleaf[(int64_t)row * v.n + pos] = u;
            if (xs) {
                const int tid = v.leaf_tid[(int64_t)v.tree(row) * v.n + pos];
                float x = (float)u;                           // xs[output_idx] = t.us[i]
                x = (float)fmax((double)x, leaf_floor);       // max(xs[..], 1e-16) in f64
                if (clamp_lo > 0.0f) x = fminf(fmaxf(x, clamp_lo), clamp_hi);
                xs[(int64_t)row * xs_rs + (int64_t)tid * xs_es] = x;
                if (efflens) p0 = (double)(x / efflens[(int64_t)row * efflens_rs + tid]);
            }
        } else if (type == TOUR_ENTER) {
            if (logu) logu[(int64_t)row * (v.n - 1) + v.tour_tgt[tb + e]] = incl;
            p1 = incl;  // ladj = sum over internal nodes of log u (ptt.jl:150-152)
        }
    }
};

// Prefix over leaf order, stored as C[row][0..n] (exclusive prefix, C[n] = total).
struct LeafPrefixEmit {
    int32_t n;
    dd *C;
    __device__ void operator()(int row, int64_t pos, dd excl, dd incl, double &p0, double &p1) const
    {
        C[(int64_t)row * (n + 1) + pos] = excl;
        if (pos == n - 1) C[(int64_t)row * (n + 1) + n] = incl;
        p0 = 0.0;
        p1 = 0.0;
    }
};

// Scan with per-chunk partial sums of two side values produced by the emit functor.
template <typename T, typename Load, typename Emit>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_partial_kernel(Load load, Emit emit, int64_t len,
                                                                         int nchunks, const T *chunk_offsets,
                                                                         double *partials /* [rows][nchunks][2] or null */)
{
    __shared__ T smem[SCAN_THREADS / 64];
    __shared__ double smd[SCAN_THREADS / 64];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const int64_t base = (int64_t)chunk * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T acc = ScanOps<T>::zero();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < len) ? load(row, base + j) : ScanOps<T>::zero();
        acc = ScanOps<T>::add(acc, v[j]);
    }
    T tot;
    T off = block_exclusive_scan<T>(acc, smem, &tot);
    off = ScanOps<T>::add(chunk_offsets[(int64_t)row * nchunks + chunk], off);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        T inc = ScanOps<T>::add(off, v[j]);
        if (base + j < len) {
            double p0, p1;
            emit(row, base + j, off, inc, p0, p1);
            s0 += p0;
            s1 += p1;
        }
        off = inc;
    }
    if (partials) {
        s0 = block_sum_f64(s0, smd);
        s1 = block_sum_f64(s1, smd);
        if (threadIdx.x == 0) {
            partials[((int64_t)row * nchunks + chunk) * 2 + 0] = s0;
            partials[((int64_t)row * nchunks + chunk) * 2 + 1] = s1;
        }
    }
}

template <typename T, typename Load, typename Emit>
inline hipError_t run_scan_partial(hipStream_t stream, int rows, int64_t len, T *chunk_buf, double *partials,
                                   Load load, Emit emit)
{
    if (rows <= 0 || len <= 0) return hipSuccess;
    const int nchunks = scan_num_chunks(len);
    dim3 grid(nchunks, rows);
    if (nchunks > 1) {
        hipLaunchKernelGGL((scan_reduce_kernel<T, Load>), grid, dim3(SCAN_THREADS), 0, stream, load, len, nchunks,
                           chunk_buf);
        hipLaunchKernelGGL((scan_spine_kernel<T>), dim3(rows), dim3(SCAN_THREADS), 0, stream, chunk_buf, nchunks);
    } else {
        hipError_t e = hipMemsetAsync(chunk_buf, 0, sizeof(T) * rows, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((scan_apply_partial_kernel<T, Load, Emit>), grid, dim3(SCAN_THREADS), 0, stream, load, emit,
                       len, nchunks, chunk_buf, partials);
    return hipGetLastError();
}

// The open-edge lists of an Euler tour cut into chunks of `chunk` entries: for every chunk the ENTER codes still open at its
// first entry = the path from the root to that point.  A chunk's offset in the tour's prefix is the sum of their values (every
// ENTER whose EXIT lies before the chunk has cancelled), so a scan that has the lists needs no reduce launch and no spine:
// the VI loop's forward kernel (vi.hip) and the approximation density's gradient scan (approx.hip) use them.  optr gets
// nchunks + 1 offsets into ocode (appended to; offsets are absolute positions in ocode).  false when the lists would pass
// `limit` entries (a caterpillar tree of 200 000 leaves needs n^2 / chunk of them) or the tour is malformed.
inline bool build_open_lists(const uint32_t *code, int64_t TL, int chunk, size_t limit, std::vector<uint32_t> &optr,
                             std::vector<uint32_t> &ocode)
{
    std::vector<uint32_t> stack;
    for (int64_t e = 0; e < TL; ++e) {
        if (e % chunk == 0) {
            optr.push_back((uint32_t)ocode.size());
            ocode.insert(ocode.end(), stack.begin(), stack.end());
            if (ocode.size() > limit) return false;
        }
        const uint32_t type = code[(size_t)e] & 3u;
        if (type == TOUR_ENTER) stack.push_back(code[(size_t)e]);
        else if (type == TOUR_EXIT) {
            if (stack.empty()) return false;
            stack.pop_back();
        }
    }
    optr.push_back((uint32_t)ocode.size());
    return true;
}

// scan_apply_partial_kernel with the chunk offsets taken from open-edge lists (build_open_lists with chunk = SCAN_CHUNK):
// Load must offer `double term(int row, uint32_t code)`, the value of an ENTER entry.  open_ptr is [trees][nchunks + 1] (a
// tree's offsets are absolute positions in open_code), shared by the rows of a tree.
template <typename Load, typename Emit>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_partial_open_kernel(Load load, Emit emit, int64_t len, int nchunks,
                                                                              const uint32_t *__restrict__ open_ptr,
                                                                              const uint32_t *__restrict__ open_code, int tree_of_row_stride,
                                                                              double *partials /* [rows][nchunks][2] or null */)
{
    __shared__ dd smem[SCAN_THREADS / 64];
    __shared__ double smd[SCAN_THREADS / 64];
    __shared__ dd spre;
    const int row = blockIdx.y, chunk = blockIdx.x;
    const int64_t base = (int64_t)chunk * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    // (the first wave sums the open edges' values, a double-double shuffle tree, and leaves the sum in LDS in front of the
    // block scan's barriers)
    if (threadIdx.x < 64) {
        const uint32_t *op = open_ptr + (size_t)(row * tree_of_row_stride) * (nchunks + 1);
        const uint32_t ob = op[chunk], oe = op[chunk + 1];
        dd pre{0.0, 0.0};
        for (uint32_t e = ob + threadIdx.x; e < oe; e += 64) pre = dd_add(pre, dd_make(load.term(row, open_code[e])));
        pre = wave_inclusive_scan<dd>(pre);  // (DPP: the wave's sum lands in its last lane)
        if (threadIdx.x == 63) spre = pre;
    }
    dd v[SCAN_ITEMS];
    dd acc{0.0, 0.0};
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < len) ? load(row, base + j) : dd{0.0, 0.0};
        acc = dd_add(acc, v[j]);
    }
    dd tot;
    dd off = block_exclusive_scan<dd>(acc, smem, &tot);
    off = dd_add(spre, off);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        dd inc = dd_add(off, v[j]);
        if (base + j < len) {
            double p0, p1;
            emit(row, base + j, off, inc, p0, p1);
            s0 += p0;
            s1 += p1;
        }
        off = inc;
    }
    if (partials) {
        s0 = block_sum_f64(s0, smd);
        s1 = block_sum_f64(s1, smd);
        if (threadIdx.x == 0) {
            partials[((int64_t)row * nchunks + chunk) * 2 + 0] = s0;
            partials[((int64_t)row * nchunks + chunk) * 2 + 1] = s1;
        }
    }
}

// Sums the per-chunk partials of each row: out[row][0..1].  One workgroup per row.
__global__ void reduce_partials_kernel(const double *partials, int nchunks, double *out, int out_stride);

// Device-level entry points (all pointers are device pointers; work is enqueued on the
// context's stream; nothing synchronises).
struct FwdOut {
    double *uleaf = nullptr;
    double *logu = nullptr;
    float *xs = nullptr;
    int64_t xs_rs = 0, xs_es = 1;
    double leaf_floor = 1e-16;
    float clamp_lo = 0.0f, clamp_hi = 1.0f;
    const float *efflens = nullptr;
    int64_t efflens_rs = 0;
    const double *ly = nullptr, *l1y = nullptr;  // optional precomputed log y / log1p(-y), [B][n-1]
    double *row_sums = nullptr;  // [B][2]: {sum x/efflen, ladj}; needs partial reduction
};
polee_status ptt_forward_device(polee_ptt *t, const double *d_ys, int32_t B, const FwdOut &out);
// One handle holding T trees over the same n (row b of a batch uses tree b); index arrays are [T][N].
void ptt_retain(polee_ptt *t);
void ptt_release(polee_ptt *t);
polee_status ptt_create_multi(polee_ctx *ctx, const int32_t *left_index, const int32_t *right_index,
                              const int32_t *leaf_index, int32_t T, int32_t N, polee_ptt **out);

}  // namespace polee
