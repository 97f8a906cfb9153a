// Polya tree transform on gfx950: plan construction (host) and the standalone tree API.
// Replaces src/ptt.jl:89-309 and src/tensorflow_ext/hsb_ops.cpp (HSB, InvHSB, InvHSBGrad).
#include "ptt_internal.hpp"

#include <algorithm>

namespace polee {

// src/ptt.jl:89-116 (first-seen child of a parent is its RIGHT child) and :293-309.
std::string children_from_parents(const int32_t *parent, const int32_t *js, int32_t N, std::vector<int32_t> &left,
                                  std::vector<int32_t> &right, std::vector<int32_t> &leaf)
{
    left.assign(N, -1);
    right.assign(N, -1);
    leaf.assign(N, -1);
    for (int32_t i = 0; i < N; ++i) {
        leaf[i] = js[i] - 1;
        const int32_t p = parent[i];
        if (i == 0) {
            if (p != 0) return "node 1 must be the root (parent 0)";
            continue;
        }
        if (p < 1 || p > N) return "node_parent_idxs out of range";
        if (p - 1 >= i) return "a parent must precede its children in node order";
        if (right[p - 1] == -1)
            right[p - 1] = i;
        else if (left[p - 1] == -1)
            left[p - 1] = i;
        else
            return "a node has more than two children";
    }
    return "";
}

std::string build_ptt_plan(const int32_t *left, const int32_t *right, const int32_t *leaf, int32_t N, PttPlan &pl)
{
    if (N < 1 || (N & 1) == 0) return "number of nodes must be odd and positive";
    const int32_t n = (N + 1) / 2;
    pl.n = n;
    pl.N = N;
    pl.TL = 3 * (int64_t)n - 2;
    pl.left.assign(left, left + N);
    pl.right.assign(right, right + N);
    pl.leaf.assign(leaf, leaf + N);
    pl.node_k.assign(N, -1);
    int32_t k = 0, nleaf = 0;
    for (int32_t i = 0; i < N; ++i) {
        const bool is_leaf = leaf[i] >= 0;
        if (is_leaf) {
            if (leaf[i] >= n) return "leaf index out of range";
            if (left[i] >= 0 || right[i] >= 0) return "a leaf has children";
            ++nleaf;
        } else {
            if (left[i] < 0 || right[i] < 0 || left[i] >= N || right[i] >= N || left[i] == right[i])
                return "an internal node lacks a child";
            if (left[i] <= i || right[i] <= i) return "a parent must precede its children in node order";
            pl.node_k[i] = k++;
        }
    }
    if (nleaf != n || k != n - 1) return "tree is not a full binary tree over n leaves";

    pl.tour_code.assign(pl.TL, 0);
    pl.tour_tgt.assign(pl.TL, 0);
    pl.leaf_tid.assign(n, -1);
    pl.tid_pos.assign(n, -1);
    pl.lo.assign(std::max(n - 1, 0), 0);
    pl.mid.assign(std::max(n - 1, 0), 0);
    pl.hi1.assign(std::max(n - 1, 0), 0);

    struct Frame {
        int32_t node;
        uint32_t edge;  // (kpar << 4) | side << 3 | root << 2
        int32_t state;
        int32_t depth;
    };
    std::vector<Frame> stack;
    stack.push_back({0, 4u, 0, 0});
    int64_t e = 0;
    int32_t pos = 0, visited = 0;
    std::vector<uint8_t> seen(N, 0);
    while (!stack.empty()) {
        Frame &f = stack.back();
        const int32_t i = f.node;
        if (leaf[i] >= 0) {
            if (seen[i]) return "a node is reachable twice";
            seen[i] = 1;
            ++visited;
            if (pl.tid_pos[leaf[i]] != -1) return "a transcript appears in two leaves";
            pl.tid_pos[leaf[i]] = pos;
            pl.leaf_tid[pos] = leaf[i];
            if (e >= pl.TL) return "tour overflow";
            pl.tour_code[e] = f.edge | TOUR_LEAF;
            pl.tour_tgt[e] = pos;
            ++e;
            ++pos;
            pl.max_depth = std::max(pl.max_depth, f.depth);
            stack.pop_back();
            continue;
        }
        const int32_t kk = pl.node_k[i];
        if (f.state == 0) {
            if (seen[i]) return "a node is reachable twice";
            seen[i] = 1;
            ++visited;
            if (e >= pl.TL) return "tour overflow";
            pl.tour_code[e] = f.edge | TOUR_ENTER;
            pl.tour_tgt[e] = kk;
            ++e;
            pl.lo[kk] = pos;
            f.state = 1;
            const int32_t d = f.depth;
            stack.push_back({right[i], ((uint32_t)kk << 4), 0, d + 1});  // right first, edge factor (1-y)
        } else if (f.state == 1) {
            pl.mid[kk] = pos;
            f.state = 2;
            const int32_t d = f.depth;
            stack.push_back({left[i], ((uint32_t)kk << 4) | 8u, 0, d + 1});  // left, edge factor y
        } else {
            pl.hi1[kk] = pos;
            if (e >= pl.TL) return "tour overflow";
            pl.tour_code[e] = f.edge | TOUR_EXIT;
            pl.tour_tgt[e] = kk;
            ++e;
            stack.pop_back();
        }
    }
    if (visited != N || e != pl.TL || pos != n) return "tree is not connected";
    return "";
}

__global__ void reduce_partials_kernel(const double *partials, int nchunks, double *out, int out_stride)
{
    __shared__ double smd[SCAN_THREADS / 64];
    const int row = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int c = threadIdx.x; c < nchunks; c += blockDim.x) {
        s0 += partials[((int64_t)row * nchunks + c) * 2 + 0];
        s1 += partials[((int64_t)row * nchunks + c) * 2 + 1];
    }
    s0 = block_sum_f64(s0, smd);
    s1 = block_sum_f64(s1, smd);
    if (threadIdx.x == 0) {
        out[(int64_t)row * out_stride + 0] = s0;
        out[(int64_t)row * out_stride + 1] = s1;
    }
}

polee_status ptt_forward_device(polee_ptt *t, const double *d_ys, int32_t B, const FwdOut &o)
{
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(t->reserve(B));
    const EdgeLogs el{d_ys, o.ly, o.l1y, (int64_t)t->n - 1};
    FwdLoad load{t->view(), el};
    FwdEmit emit{t->view(), el,         o.uleaf,    o.logu,     o.xs,     o.xs_rs,
                 o.xs_es,   o.leaf_floor, o.clamp_lo, o.clamp_hi, o.efflens, o.efflens_rs};
    double *partials = o.row_sums ? t->d_part.p : nullptr;
    hipError_t e = run_scan_partial<double>(ctx->stream, B, t->TL, (double *)t->d_chunk.p, partials, load, emit);
    if (e != hipSuccess) return fail(ctx, POLEE_ERR_HIP, "forward scan launch failed: %s", hipGetErrorString(e));
    if (o.row_sums) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(B), dim3(SCAN_THREADS), 0, ctx->stream, t->d_part.p,
                           scan_num_chunks(t->TL), o.row_sums, 2);
        POLEE_KERNEL_CHECK(ctx);
    }
    return POLEE_OK;
}

// ---- node-level kernels of the standalone API ---------------------------------------------

// Leaf-order load of x (f32, row-major [B][n] by transcript id) as double-double.
struct LeafLoadF32 {
    PttView v;
    const float *x;
    __device__ dd operator()(int row, int64_t pos) const
    {
        const int tid = v.leaf_tid[(int64_t)v.tree(row) * v.n + pos];
        return dd_make((double)x[(int64_t)row * v.n + tid]);
    }
};

// a = u_leaf * x_grad, the summand of the subtree sums of transform_gradients!.
struct LeafLoadGradF64 {
    PttView v;
    const double *uleaf;   // [B][n] leaf order
    const double *x_grad;  // [B][n] transcript order
    __device__ dd operator()(int row, int64_t pos) const
    {
        const int tid = v.leaf_tid[(int64_t)v.tree(row) * v.n + pos];
        return dd_make(uleaf[(int64_t)row * v.n + pos] * x_grad[(int64_t)row * v.n + tid]);
    }
};

// transform_gradients! (src/ptt.jl:167-209) in closed form.  With H_i = u_i (g1_i + g2_i)
// the reference recursion gives H_i = [i internal] + H_left + H_right, H_leaf = u x_grad,
// hence H_i = (#internal nodes in subtree i) + sum_{leaves} u x_grad, and
// y_grad[k] = u_i ((g1_l+g2_l) - (g1_r+g2_r)) = H_l / y - H_r / (1 - y).
__global__ void ptt_grad_nodes_kernel(PttView v, const double *ys, const dd *C, int with_ladj, double *y_grad)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (k >= v.n - 1) return;
    const int64_t tb = (int64_t)v.tree(row) * (v.n - 1);
    const int lo = v.lo[tb + k], mid = v.mid[tb + k], hi1 = v.hi1[tb + k];
    const dd *Cr = C + (int64_t)row * (v.n + 1);
    double Hr = dd_diff(Cr[mid], Cr[lo]);
    double Hl = dd_diff(Cr[hi1], Cr[mid]);
    if (with_ladj) {
        Hr += (double)(mid - lo - 1);
        Hl += (double)(hi1 - mid - 1);
    }
    const double y = ys[(int64_t)row * (v.n - 1) + k];
    y_grad[(int64_t)row * (v.n - 1) + k] = Hl / y - Hr / (1.0 - y);
}

// inverse_transform! (src/ptt.jl:257-285) / InvHSB (hsb_ops.cpp:212-238): subtree sums
// from the double-double leaf prefix; y = u_left / u; ladj -= log u.
// julia_log != 0 takes the log of Float32(u) as ptt.jl:277 does.
__global__ void ptt_inverse_nodes_kernel(PttView v, const dd *C, int julia_log, double *ys, double *ladj)
{
    __shared__ double smd[SCAN_THREADS / 64];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    double contrib = 0.0;
    if (k < v.n - 1) {
        const int64_t tb = (int64_t)v.tree(row) * (v.n - 1);
        const int lo = v.lo[tb + k], mid = v.mid[tb + k], hi1 = v.hi1[tb + k];
        const dd *Cr = C + (int64_t)row * (v.n + 1);
        const double ur = dd_diff(Cr[mid], Cr[lo]);
        const double ul = dd_diff(Cr[hi1], Cr[mid]);
        const double u = ul + ur;
        ys[(int64_t)row * (v.n - 1) + k] = ul / u;
        contrib = julia_log ? -(double)logf((float)u) : -log(u);
    }
    contrib = block_sum_f64(contrib, smd);
    if (threadIdx.x == 0 && ladj) atomicAdd(&ladj[row], contrib);
}

__global__ void sum_rows_kernel(const double *v, int64_t len, double *out)
{
    __shared__ double smd[SCAN_THREADS / 64];
    const int row = blockIdx.y;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x)
        s += v[(int64_t)row * len + i];
    s = block_sum_f64(s, smd);
    if (threadIdx.x == 0) atomicAdd(&out[row], s);
}

__global__ void logistic_f32_to_f64_kernel(const float *logit, int64_t len, double *y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) y[i] = 1.0 / (1.0 + (double)expf(-logit[i]));  // hsb_ops.cpp:103
}

// InvHSBGrad (hsb_ops.cpp:342-391): v_child = v_parent - ladj_grad/u_j +- (u_other/u_j^2) y_grad
// is an additive recursion down the tree -> Euler-tour scan of the edge terms.  Terms carry
// 1/u_j (huge for tiny subtrees) and are subtracted again on EXIT, so the scan runs in
// double-double.
struct InvGradLoad {
    PttView v;
    const double *y, *y_grad, *logu;  // [B][n-1]
    const float *ladj_grad;           // [B]
    __device__ inline double term(int row, uint32_t code) const
    {
        if (code & 4u) return 0.0;
        const int64_t o = (int64_t)row * (v.n - 1) + (code >> 4);
        const double inv_u = exp(-logu[o]);
        const double yy = y[o];
        const double w = (code & 8u) ? (1.0 - yy) : -yy;  // u_right/u_j^2 = (1-y)/u_j ; -u_left/u_j^2 = -y/u_j
        return -(double)ladj_grad[row] * inv_u + w * inv_u * y_grad[o];
    }
    __device__ dd operator()(int row, int64_t e) const
    {
        const uint32_t code = v.tour_code[(int64_t)v.tree(row) * v.TL + e];
        const uint32_t type = code & 3u;
        if (type == TOUR_LEAF) return dd_make(0.0);
        const double t = term(row, code);
        return dd_make(type == TOUR_ENTER ? t : -t);
    }
};
struct InvGradEmit {
    InvGradLoad l;
    float *backprops;  // [B][n] transcript order
    __device__ void operator()(int row, int64_t e, dd /*excl*/, dd incl, double &p0, double &p1) const
    {
        p0 = p1 = 0.0;
        const int64_t tb = (int64_t)l.v.tree(row) * l.v.TL;
        const uint32_t code = l.v.tour_code[tb + e];
        if ((code & 3u) != TOUR_LEAF) return;
        const dd tot = dd_add(incl, dd_make(l.term(row, code)));
        const int pos = l.v.tour_tgt[tb + e];
        const int tid = l.v.leaf_tid[(int64_t)l.v.tree(row) * l.v.n + pos];
        backprops[(int64_t)row * l.v.n + tid] = (float)(tot.hi + tot.lo);
    }
};

}  // namespace polee

using namespace polee;

polee_status polee_ptt::reserve(int32_t rows)
{
    if (rows <= cap_rows) return POLEE_OK;
    const size_t R = rows;
    const int nch = std::max(scan_num_chunks(TL), 1);
    POLEE_TRY(d_chunk.alloc(ctx, R * nch));
    POLEE_TRY(d_ys.alloc(ctx, R * std::max(n - 1, 1)));
    POLEE_TRY(d_uleaf.alloc(ctx, R * n));
    POLEE_TRY(d_logu.alloc(ctx, R * std::max(n - 1, 1)));
    POLEE_TRY(d_C.alloc(ctx, R * (n + 1)));
    POLEE_TRY(d_part.alloc(ctx, R * nch * 2));
    POLEE_TRY(d_row.alloc(ctx, R * 4));
    cap_rows = rows;
    return POLEE_OK;
}

static polee_status ptt_create_from_plans(polee_ctx *ctx, std::vector<PttPlan> &&plans, polee_ptt **out)
{
    polee_ptt *t = new (std::nothrow) polee_ptt();
    if (!t) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    t->ctx = ctx;
    ctx_retain(ctx);
    t->plans = std::move(plans);
    t->T = (int32_t)t->plans.size();
    t->n = t->plans[0].n;
    t->N = t->plans[0].N;
    t->TL = t->plans[0].TL;
    std::vector<uint32_t> code;
    std::vector<int32_t> tgt, ltid, lo, mid, hi1;
    for (auto &p : t->plans) {
        code.insert(code.end(), p.tour_code.begin(), p.tour_code.end());
        tgt.insert(tgt.end(), p.tour_tgt.begin(), p.tour_tgt.end());
        ltid.insert(ltid.end(), p.leaf_tid.begin(), p.leaf_tid.end());
        lo.insert(lo.end(), p.lo.begin(), p.lo.end());
        mid.insert(mid.end(), p.mid.begin(), p.mid.end());
        hi1.insert(hi1.end(), p.hi1.begin(), p.hi1.end());
    }
    if (lo.empty()) {  // n == 1: keep the buffers non-null
        lo.push_back(0);
        mid.push_back(0);
        hi1.push_back(0);
    }
    polee_status s;
    if ((s = t->d_tour_code.upload(ctx, code)) || (s = t->d_tour_tgt.upload(ctx, tgt)) ||
        (s = t->d_leaf_tid.upload(ctx, ltid)) || (s = t->d_lo.upload(ctx, lo)) || (s = t->d_mid.upload(ctx, mid)) ||
        (s = t->d_hi1.upload(ctx, hi1))) {
        ptt_release(t);
        return s;
    }
    *out = t;
    return POLEE_OK;
}

namespace polee {
void ptt_retain(polee_ptt *t)
{
    if (t) ++t->refs;
}
void ptt_release(polee_ptt *t)
{
    if (!t || --t->refs > 0) return;
    polee_ctx *ctx = t->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    delete t;
    ctx_release(ctx);
}
polee_status ptt_create_multi(polee_ctx *ctx, const int32_t *left_index, const int32_t *right_index,
                              const int32_t *leaf_index, int32_t T, int32_t N, polee_ptt **out)
{
    if (T < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "need at least one tree");
    std::vector<PttPlan> plans(T);
    for (int32_t s = 0; s < T; ++s) {
        std::string err = build_ptt_plan(left_index + (size_t)s * N, right_index + (size_t)s * N,
                                         leaf_index + (size_t)s * N, N, plans[s]);
        if (!err.empty()) return fail(ctx, POLEE_ERR_BAD_ARG, "malformed tree %d: %s", s, err.c_str());
    }
    return ptt_create_from_plans(ctx, std::move(plans), out);
}
}  // namespace polee

extern "C" {

polee_status polee_make_inverse_ptt_params(const int32_t *node_parent_idxs, const int32_t *node_js, int32_t N,
                                           int32_t *left_index, int32_t *right_index, int32_t *leaf_index)
{
    if (!node_parent_idxs || !node_js || !left_index || !right_index || !leaf_index || N < 1)
        return fail(nullptr, POLEE_ERR_BAD_ARG, "polee_make_inverse_ptt_params: bad argument");
    std::vector<int32_t> l, r, f;
    std::string err = children_from_parents(node_parent_idxs, node_js, N, l, r, f);
    if (!err.empty()) return fail(nullptr, POLEE_ERR_BAD_ARG, "malformed tree: %s", err.c_str());
    std::copy(l.begin(), l.end(), left_index);
    std::copy(r.begin(), r.end(), right_index);
    std::copy(f.begin(), f.end(), leaf_index);
    return POLEE_OK;
}

polee_status polee_ptt_create_from_index(polee_ctx *ctx, const int32_t *left_index, const int32_t *right_index,
                                         const int32_t *leaf_index, int32_t N, polee_ptt **out)
{
    POLEE_TRY(use_device(ctx));
    if (!left_index || !right_index || !leaf_index || !out) return fail(ctx, POLEE_ERR_BAD_ARG, "null argument");
    std::vector<PttPlan> plans(1);
    std::string err = build_ptt_plan(left_index, right_index, leaf_index, N, plans[0]);
    if (!err.empty()) return fail(ctx, POLEE_ERR_BAD_ARG, "malformed tree: %s", err.c_str());
    return ptt_create_from_plans(ctx, std::move(plans), out);
}

polee_status polee_ptt_create(polee_ctx *ctx, const int32_t *node_parent_idxs, const int32_t *node_js, int32_t N,
                              polee_ptt **out)
{
    POLEE_TRY(use_device(ctx));
    if (!node_parent_idxs || !node_js || !out || N < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "null argument");
    std::vector<int32_t> l, r, f;
    std::string err = children_from_parents(node_parent_idxs, node_js, N, l, r, f);
    if (!err.empty()) return fail(ctx, POLEE_ERR_BAD_ARG, "malformed tree: %s", err.c_str());
    return polee_ptt_create_from_index(ctx, l.data(), r.data(), f.data(), N, out);
}

void polee_ptt_destroy(polee_ptt *t) { ptt_release(t); }

int32_t polee_ptt_n(const polee_ptt *t) { return t ? t->n : 0; }

polee_status polee_ptt_transform(polee_ptt *t, const double *ys, int32_t B, float *xs, double *ladj)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!ys || !xs || B < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_ptt_transform: bad argument");
    POLEE_TRY(t->reserve(B));
    const size_t nm1 = t->n - 1;
    POLEE_TRY(t->d_ys.upload(ctx, ys, (size_t)B * nm1));
    POLEE_TRY(t->d_f32a.alloc(ctx, (size_t)B * t->n));
    FwdOut o;
    o.uleaf = t->d_uleaf.p;
    o.logu = t->d_logu.p;
    o.xs = t->d_f32a.p;
    o.xs_rs = t->n;
    o.row_sums = ladj ? t->d_row.p : nullptr;
    POLEE_TRY(ptt_forward_device(t, t->d_ys.p, B, o));
    POLEE_TRY(t->d_f32a.download(ctx, xs, (size_t)B * t->n));
    if (ladj) {
        std::vector<double> rs((size_t)B * 2);
        POLEE_TRY(t->d_row.download(ctx, rs.data(), rs.size()));
        for (int b = 0; b < B; ++b) {
            ladj[b] = rs[(size_t)b * 2 + 1];
            if (!std::isfinite(ladj[b]))
                return fail(ctx, POLEE_ERR_NONFINITE, "transform!: non-finite ladj (ptt.jl:157)");
        }
    }
    return POLEE_OK;
}

polee_status polee_ptt_transform_gradients(polee_ptt *t, const double *ys, const double *x_grad, int32_t B,
                                           int with_ladj, double *y_grad)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!ys || !x_grad || !y_grad || B < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (t->n < 2) return POLEE_OK;
    POLEE_TRY(t->reserve(B));
    const size_t nm1 = t->n - 1, n = t->n;
    POLEE_TRY(t->d_ys.upload(ctx, ys, (size_t)B * nm1));
    POLEE_TRY(t->d_f64a.upload(ctx, x_grad, (size_t)B * n));
    POLEE_TRY(t->d_f64b.alloc(ctx, (size_t)B * nm1));
    // u of the leaves (the reference reuses t.us of the preceding transform!; recomputed here)
    FwdOut o;
    o.uleaf = t->d_uleaf.p;
    POLEE_TRY(ptt_forward_device(t, t->d_ys.p, B, o));
    LeafLoadGradF64 load{t->view(), t->d_uleaf.p, t->d_f64a.p};
    LeafPrefixEmit emit{t->n, t->d_C.p};
    hipError_t e = run_scan_partial<dd>(ctx->stream, B, t->n, t->d_chunk.p, nullptr, load, emit);
    if (e != hipSuccess) return fail(ctx, POLEE_ERR_HIP, "scan launch failed: %s", hipGetErrorString(e));
    dim3 grid((unsigned)ceil_div(nm1, 256), B);
    hipLaunchKernelGGL(ptt_grad_nodes_kernel, grid, dim3(256), 0, ctx->stream, t->view(), t->d_ys.p, t->d_C.p,
                       with_ladj, t->d_f64b.p);
    POLEE_KERNEL_CHECK(ctx);
    return t->d_f64b.download(ctx, y_grad, (size_t)B * nm1);
}

static polee_status inverse_impl(polee_ptt *t, const float *xs, int32_t B, int julia_log, double *ys,
                                 double *ladj_f64, float *ladj_f32)
{
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!xs || !ys || B < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    POLEE_TRY(t->reserve(B));
    const size_t nm1 = t->n - 1, n = t->n;
    POLEE_TRY(t->d_f32a.upload(ctx, xs, (size_t)B * n));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(t->d_row.p, 0, sizeof(double) * B, ctx->stream));
    if (nm1 > 0) {
        LeafLoadF32 load{t->view(), t->d_f32a.p};
        LeafPrefixEmit emit{t->n, t->d_C.p};
        hipError_t e = run_scan_partial<dd>(ctx->stream, B, t->n, t->d_chunk.p, nullptr, load, emit);
        if (e != hipSuccess) return fail(ctx, POLEE_ERR_HIP, "scan launch failed: %s", hipGetErrorString(e));
        dim3 grid((unsigned)ceil_div(nm1, 256), B);
        hipLaunchKernelGGL(ptt_inverse_nodes_kernel, grid, dim3(256), 0, ctx->stream, t->view(), t->d_C.p, julia_log,
                           t->d_ys.p, t->d_row.p);
        POLEE_KERNEL_CHECK(ctx);
        POLEE_TRY(t->d_ys.download(ctx, ys, (size_t)B * nm1));
    }
    std::vector<double> la(B);
    POLEE_TRY(t->d_row.download(ctx, la.data(), B));
    for (int b = 0; b < B; ++b) {
        if (ladj_f64) ladj_f64[b] = la[b];
        if (ladj_f32) ladj_f32[b] = (float)la[b];
    }
    return POLEE_OK;
}

polee_status polee_ptt_inverse_transform(polee_ptt *t, const float *xs, int32_t B, double *ys, double *ladj)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    return inverse_impl(t, xs, B, 1, ys, ladj, nullptr);
}

polee_status polee_inv_hsb(polee_ptt *t, const float *x, int32_t B, double *y, float *ladj)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    return inverse_impl(t, x, B, 0, y, nullptr, ladj);
}

polee_status polee_hsb(polee_ptt *t, const float *y_logit, int32_t B, float *x)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!y_logit || !x || B < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    POLEE_TRY(t->reserve(B));
    const size_t nm1 = t->n - 1, n = t->n;
    POLEE_TRY(t->d_f32b.upload(ctx, y_logit, (size_t)B * nm1));
    POLEE_TRY(t->d_f32a.alloc(ctx, (size_t)B * n));
    if (nm1 > 0) {
        hipLaunchKernelGGL(logistic_f32_to_f64_kernel, dim3((unsigned)ceil_div(B * nm1, 256)), dim3(256), 0,
                           ctx->stream, t->d_f32b.p, (int64_t)B * nm1, t->d_ys.p);
        POLEE_KERNEL_CHECK(ctx);
    }
    FwdOut o;
    o.xs = t->d_f32a.p;
    o.xs_rs = t->n;
    o.leaf_floor = 0.0;  // the TF op does not floor (hsb_ops.cpp:99-101)
    POLEE_TRY(ptt_forward_device(t, t->d_ys.p, B, o));
    return t->d_f32a.download(ctx, x, (size_t)B * n);
}

polee_status polee_inv_hsb_grad(polee_ptt *t, const double *y_grad, const float *ladj_grad, const double *y,
                                int32_t B, float *backprops)
{
    if (!t) return fail(nullptr, POLEE_ERR_BAD_ARG, "null tree");
    polee_ctx *ctx = t->ctx;
    POLEE_TRY(use_device(ctx));
    if (!y_grad || !ladj_grad || !y || !backprops || B < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    POLEE_TRY(t->reserve(B));
    const size_t nm1 = t->n - 1, n = t->n;
    POLEE_TRY(t->d_ys.upload(ctx, y, (size_t)B * nm1));
    POLEE_TRY(t->d_f64a.upload(ctx, y_grad, (size_t)B * nm1));
    POLEE_TRY(t->d_f32b.upload(ctx, ladj_grad, (size_t)B));
    POLEE_TRY(t->d_f32a.alloc(ctx, (size_t)B * n));
    FwdOut o;
    o.logu = t->d_logu.p;
    POLEE_TRY(ptt_forward_device(t, t->d_ys.p, B, o));
    InvGradLoad load{t->view(), t->d_ys.p, t->d_f64a.p, t->d_logu.p, t->d_f32b.p};
    InvGradEmit emit{load, t->d_f32a.p};
    hipError_t e = run_scan_partial<dd>(ctx->stream, B, t->TL, t->d_chunk.p, nullptr, load, emit);
    if (e != hipSuccess) return fail(ctx, POLEE_ERR_HIP, "scan launch failed: %s", hipGetErrorString(e));
    return t->d_f32a.download(ctx, backprops, (size_t)B * n);
}

}  // extern "C"
