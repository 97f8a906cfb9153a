// Host-only introspection entry points (include/polee_hip_debug.h).
#include <algorithm>
#include <memory>

#include "../../include/polee_hip_debug.h"
#include "loglik_internal.hpp"
#include "ptt_internal.hpp"
#include "psell_device.hpp"

namespace polee {
std::string csc_to_csr(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                       const float *nzval, BVec<uint64_t> &rowptr, RawVec<uint32_t> &col, RawVec<float> &val);
}
using namespace polee;

struct polee_psell_debug {
    PsellHost h;
};

extern "C" {

polee_status polee_debug_ptt_plan(const int32_t *node_parent_idxs, const int32_t *node_js, int32_t N,
                                  uint32_t *tour_code, int32_t *tour_tgt, int32_t *leaf_tid, int32_t *lo,
                                  int32_t *mid, int32_t *hi1, int32_t *max_depth)
{
    if (!node_parent_idxs || !node_js) return fail(nullptr, POLEE_ERR_BAD_ARG, "null argument");
    std::vector<int32_t> l, r, f;
    std::string err = children_from_parents(node_parent_idxs, node_js, N, l, r, f);
    PttPlan pl;
    if (err.empty()) err = build_ptt_plan(l.data(), r.data(), f.data(), N, pl);
    if (!err.empty()) return fail(nullptr, POLEE_ERR_BAD_ARG, "malformed tree: %s", err.c_str());
    if (tour_code) std::copy(pl.tour_code.begin(), pl.tour_code.end(), tour_code);
    if (tour_tgt) std::copy(pl.tour_tgt.begin(), pl.tour_tgt.end(), tour_tgt);
    if (leaf_tid) std::copy(pl.leaf_tid.begin(), pl.leaf_tid.end(), leaf_tid);
    if (lo) std::copy(pl.lo.begin(), pl.lo.end(), lo);
    if (mid) std::copy(pl.mid.begin(), pl.mid.end(), mid);
    if (hi1) std::copy(pl.hi1.begin(), pl.hi1.end(), hi1);
    if (max_depth) *max_depth = pl.max_depth;
    return POLEE_OK;
}

static polee_status debug_psell_build_impl(int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                           const uint32_t *rowval, const float *nzval, const int64_t *ks,
                                           polee_psell_debug **out)
{
    if (!colptr || !out) return fail(nullptr, POLEE_ERR_BAD_ARG, "null argument");
    BVec<uint64_t> rowptr;
    RawVec<uint32_t> col;  // (resize leaves them uninitialised: 1.9 GB of zeros would be written by one thread)
    RawVec<float> val;
    std::string err = csc_to_csr(m, n, colptr, colptr_bytes, rowval, nzval, rowptr, col, val);
    if (!err.empty()) return fail(nullptr, POLEE_ERR_BAD_ARG, "likelihood matrix: %s", err.c_str());
    polee_psell_debug *p = new polee_psell_debug();
    err = build_psell(m, n, rowptr.data(), col.data(), val.data(), ks, p->h);
    if (!err.empty()) {
        delete p;
        return fail(nullptr, err.find("more than") != std::string::npos ? POLEE_ERR_UNSUPPORTED : POLEE_ERR_BAD_ARG,
                    "likelihood matrix: %s", err.c_str());
    }
    *out = p;
    return POLEE_OK;
}

// The layout with some of the builder's stages run on the device (bit 0: stage 1, bit 1: stage 2, bit 2: stage 3) and the rest
// on the host, each continuing from the other's output: every mix must give the bytes polee_debug_psell_build gives.
static polee_status debug_psell_build_device_impl(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                                  const uint32_t *rowval, const float *nzval, const int64_t *ks, int device_stages,
                                                  polee_psell_debug **out)
{
    if (!ctx || !colptr || !out) return fail(ctx, POLEE_ERR_BAD_ARG, "null argument");
    POLEE_TRY(use_device(ctx));
    BVec<uint64_t> rowptr;
    RawVec<uint32_t> col;
    RawVec<float> val;
    std::string err = csc_to_csr(m, n, colptr, colptr_bytes, rowval, nzval, rowptr, col, val);
    if (!err.empty()) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: %s", err.c_str());
    std::unique_ptr<polee_psell_debug> p(new polee_psell_debug());
    auto bad = [&](const std::string &e) {
        return fail(ctx, e.find("more than") != std::string::npos ? POLEE_ERR_UNSUPPORTED : POLEE_ERR_BAD_ARG, "likelihood matrix: %s", e.c_str());
    };
    DevBuf<uint64_t> d_rowptr;
    DevBuf<uint32_t> d_col;
    DevBuf<float> d_val;
    DevBuf<int64_t> d_ks;
    POLEE_TRY(d_rowptr.upload(ctx, rowptr.data(), rowptr.size()));
    POLEE_TRY(d_col.upload(ctx, col.data(), col.size()));
    POLEE_TRY(d_val.upload(ctx, val.data(), val.size()));
    if (ks) POLEE_TRY(d_ks.upload(ctx, ks, (size_t)m));
    PsellDevIn X;
    X.rowptr = d_rowptr.p; X.col = d_col.p; X.val = d_val.p; X.ks = ks ? d_ks.p : nullptr; X.m = m; X.n = n;
    PsellRuns R;
    PsellRows W;
    PsellDevRuns DR;
    PsellDevRowsOwned DW;
    bool runs_on_device = false, rows_on_device = false;
    if (device_stages & 1) {
        POLEE_TRY(psell_device_stage1(ctx, X, p->h, DR, true));
        runs_on_device = true;
    } else if ((err = psell_stage1(m, n, rowptr.data(), col.data(), val.data(), ks, p->h, R)) != "") {
        return bad(err);
    }
    if (device_stages & 2) {
        if (!runs_on_device) {
            DR.n_a1 = R.a1_rows.size(); DR.n_a2 = R.a2_rows.size(); DR.n_rb = R.rb.size();
            POLEE_TRY(DR.a1_rows.upload(ctx, R.a1_rows.data(), R.a1_rows.size()));
            POLEE_TRY(DR.a1_ends.upload(ctx, R.a1_ends.data(), R.a1_ends.size()));
            POLEE_TRY(DR.a2_rows.upload(ctx, R.a2_rows.data(), R.a2_rows.size()));
            POLEE_TRY(DR.a2_ends.upload(ctx, R.a2_ends.data(), R.a2_ends.size()));
            POLEE_TRY(DR.rb.upload(ctx, R.rb.data(), R.rb.size()));
        }
        bool needs_host = false;
        POLEE_TRY(psell_device_stage2(ctx, X, DR, p->h, DW, needs_host));
        if (needs_host) {  // (a matrix without structure: the host builder's case)
            if (runs_on_device) POLEE_TRY(psell_device_runs_to_host(ctx, DR, R));
            if ((err = psell_stage2(m, n, rowptr.data(), col.data(), val.data(), ks, R, p->h, W)) != "") return bad(err);
        } else {
            rows_on_device = true;
        }
    } else {
        if (runs_on_device) POLEE_TRY(psell_device_runs_to_host(ctx, DR, R));
        if ((err = psell_stage2(m, n, rowptr.data(), col.data(), val.data(), ks, R, p->h, W)) != "") return bad(err);
    }
    if (device_stages & 4) {
        if (!rows_on_device) {
            DW.Nr = W.rows.size();
            DW.npat = W.pat_ptr.size() - 1;
            const size_t b[7] = {0, (size_t)p->h.rows_a1, (size_t)p->h.rows_a1m, (size_t)p->h.rows_a2, (size_t)p->h.rows_a, (size_t)p->h.rows_s, W.rows.size()};
            for (int q = 0; q < 7; ++q) DW.bounds[q] = b[q];
            POLEE_TRY(DW.rows.upload(ctx, W.rows.data(), W.rows.size()));
            POLEE_TRY(DW.run_end.upload(ctx, W.run_end.data(), W.run_end.size()));
            POLEE_TRY(DW.gid.upload(ctx, W.row_gid.data(), W.row_gid.size()));
            POLEE_TRY(DW.form.upload(ctx, W.row_form.data(), W.row_form.size()));
            POLEE_TRY(DW.pat_ptr.upload(ctx, W.pat_ptr.data(), W.pat_ptr.size()));
            POLEE_TRY(DW.pat_col.upload(ctx, W.pat_col.data(), W.pat_col.size()));
        }
        PsellDevOut D;
        POLEE_TRY(psell_device_stage3(ctx, X, DW.view(), p->h, D, true));
    } else {
        if (rows_on_device) POLEE_TRY(psell_device_rows_to_host(ctx, DW, p->h, W));
        if ((err = psell_stage3(m, n, rowptr.data(), col.data(), val.data(), ks, W, p->h)) != "") return bad(err);
    }
    *out = p.release();
    return POLEE_OK;
}

// (the builders allocate through std::vector / BVec / new -- nothing may unwind through the C ABI)
polee_status polee_debug_psell_build(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                     const float *nzval, const int64_t *ks, polee_psell_debug **out)
{
    return guarded(nullptr, "polee_debug_psell_build", [&] { return debug_psell_build_impl(m, n, colptr, colptr_bytes, rowval, nzval, ks, out); });
}
polee_status polee_debug_psell_build_device(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                            const uint32_t *rowval, const float *nzval, const int64_t *ks, int device_stages,
                                            polee_psell_debug **out)
{
    return guarded(ctx, "polee_debug_psell_build_device",
                   [&] { return debug_psell_build_device_impl(ctx, m, n, colptr, colptr_bytes, rowval, nzval, ks, device_stages, out); });
}

polee_status polee_debug_psell_view(const polee_psell_debug *p, polee_psell_view *v)
{
    if (!p || !v) return fail(nullptr, POLEE_ERR_BAD_ARG, "null argument");
    const PsellHost &h = p->h;
    v->m = h.m; v->n = h.n; v->nnz = h.nnz;
    v->num_slices = h.num_slices; v->num_tiles = h.num_tiles; v->padded_nnz = h.padded_nnz;
    v->num_empty_rows = h.empty_rows;
    v->data_bytes = (int64_t)h.data.size(); v->dict_len = (int64_t)h.dict.size();
    v->max_row_nnz = h.max_row; v->max_tile_cols = h.max_tile_cols;
    v->data = h.data.data(); v->slice_off = h.slice_off.data(); v->tile_slice = h.tile_slice.data();
    v->tile_dict = h.tile_dict.data(); v->dict = h.dict.data(); v->row_order = h.row_order.data();
    v->slice_ks = h.slice_ks.empty() ? nullptr : h.slice_ks.data();
    v->slice_flags = h.slice_flags.data();
    v->num_tiles_a = h.num_tiles_a;
    v->num_tiles_a1 = h.num_tiles_a1;
    v->num_tiles_a1m = h.num_tiles_a1m;
    v->num_tiles_a2 = h.num_tiles_a2;
    v->num_tiles_s = h.num_tiles_s;
    v->slice_w = h.slice_w.data();
    for (int i = 0; i < 8; ++i) v->stream_rows[i] = v->stream_nnz[i] = v->stream_bytes[i] = 0;
    v->csr_num_rows = (int64_t)h.csr_rows.size();
    v->csr_rowptr = h.csr_rowptr.data();
    v->csr_col = h.csr_col.data();
    v->csr_val = h.csr_val.data();
    v->csr_rows = h.csr_rows.data();
    v->single_num_rows = (int64_t)h.single_rows.size();
    v->single_rows = h.single_rows.data();
    v->single_cnt = h.single_cnt.empty() ? nullptr : h.single_cnt.data();
    v->single_logsum = h.single_logsum;
    for (int i = 0; i < PSELL_NSTREAMS; ++i) {
        v->stream_rows[i] = h.stream_rows[i];
        v->stream_nnz[i] = h.stream_nnz[i];
        v->stream_bytes[i] = h.stream_bytes[i];
    }
    return POLEE_OK;
}

void polee_debug_psell_free(polee_psell_debug *p) { delete p; }

}  // extern "C"
