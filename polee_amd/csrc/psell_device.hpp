// Device-side layout builder (psell_device.hip): interfaces of its stages, mirroring psell_build.cpp's.
#pragma once
#include "loglik_internal.hpp"

namespace polee {

struct PsellDevIn {  // X by rows (CSR, 0-based) in device memory
    const uint64_t *rowptr = nullptr;
    const uint32_t *col = nullptr;
    const float *val = nullptr;
    const int64_t *ks = nullptr;  // or null
    int64_t m = 0, n = 0;
};
struct PsellDevCSR {  // X by rows, owned (val / ks may be borrowed from the caller's device arrays)
    DevBuf<uint64_t> rowptr;
    DevBuf<uint32_t> col;
    DevBuf<float> val;
    DevBuf<int64_t> ks;
    const float *val_ptr = nullptr;
    const int64_t *ks_ptr = nullptr;
    int64_t m = 0, n = 0;
    PsellDevIn view() const
    {
        PsellDevIn v;
        v.rowptr = rowptr.p; v.col = col.p; v.val = val_ptr; v.ks = ks_ptr; v.m = m; v.n = n;
        return v;
    }
};
struct PsellDevRows {  // PsellRows in device memory
    const uint32_t *rows = nullptr, *run_end = nullptr, *gid = nullptr;
    const uint8_t *form = nullptr;
    const uint32_t *pat_ptr = nullptr, *pat_col = nullptr;
    size_t Nr = 0;
    size_t bounds[7] = {};  // rows of stream st: [bounds[st], bounds[st + 1])
};
struct PsellDevOut {  // what stays on the device
    DevBuf<uint8_t> data;  // the slice stream (+ 2 KiB of slack)
    size_t data_bytes = 0;
    DevBuf<float> slice_ks;
};

struct PsellDevRuns {  // PsellRuns in device memory (+ stream S's counts)
    DevBuf<uint32_t> a1_rows, a1_ends, a2_rows, a2_ends, rb;
    size_t n_a1 = 0, n_a2 = 0, n_rb = 0;
    DevBuf<float> single_cnt;  // [n], allocated when there are such rows
};

// stage 1 on the device: keys, stable sort, runs.  Fills `out` as psell_stage1 does (single_rows only with want_debug).
polee_status psell_device_stage1(polee_ctx *ctx, const PsellDevIn &X, PsellHost &out, PsellDevRuns &R, bool want_debug);
polee_status psell_device_runs_to_host(polee_ctx *ctx, const PsellDevRuns &R, PsellRuns &H);

struct PsellDevRowsOwned {  // stage 2's result in device memory
    DevBuf<uint32_t> rows, run_end, gid, pat_ptr, pat_col;
    DevBuf<uint8_t> form;
    size_t Nr = 0, npat = 0;
    size_t bounds[7] = {};
    PsellDevRows view() const
    {
        PsellDevRows v;
        v.rows = rows.p; v.run_end = run_end.p; v.gid = gid.p; v.form = form.p; v.pat_ptr = pat_ptr.p; v.pat_col = pat_col.p;
        v.Nr = Nr;
        for (int q = 0; q < 7; ++q) v.bounds[q] = bounds[q];
        return v;
    }
};

// stage 2 on the device: packing of the leftover rows, mixed streams, the ordered rows.  needs_host: the matrix has a real share
// of rows without any structure (stream C's question, sequential over all of them): the layout is the host builder's to make.
polee_status psell_device_stage2(polee_ctx *ctx, const PsellDevIn &X, PsellDevRuns &R, PsellHost &out, PsellDevRowsOwned &W, bool &needs_host);
polee_status psell_device_rows_to_host(polee_ctx *ctx, const PsellDevRowsOwned &W, PsellHost &out, PsellRows &H);

// stage 3 on the device: slices and tiles.  Fills `out`'s metadata (offsets, tiles, dictionaries, flags, totals); the bytes stay
// in D -- and are copied into out.data / row_order / slice_ks as well with want_debug.
polee_status psell_device_stage3(polee_ctx *ctx, const PsellDevIn &X, const PsellDevRows &W, PsellHost &out, PsellDevOut &D,
                                 bool want_debug);

// the input onto the device, by rows: from X by columns (host arrays, 1-based: polee_loglik_create's arguments; a stable sort by
// row on the device) or from Xt (1-based; host arrays, or device arrays: an xbuild result)
polee_status psell_device_rows_from_csc(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                        const float *nzval, const int64_t *ks, PsellDevCSR &C, bool &needs_host);
// ... from X by columns already in device memory (polee_devx: one copy for the tree and the layout)
polee_status psell_device_rows_from_dev_csc(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *d_cp, uint64_t nnz, const uint32_t *d_rowval,
                                            const float *d_nzval, const int64_t *ks, PsellDevCSR &C, DevBuf<uint32_t> *own_rowval,
                                            const float *late_nzval = nullptr, DevBuf<float> *late_buf = nullptr);  // (host values, uploaded beside the sort)
polee_status psell_check_colptr(polee_ctx *ctx, int64_t n, const void *colptr, int colptr_bytes, std::vector<uint64_t> &cp, uint64_t &nnz);
polee_status psell_device_rows_from_xt(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *tcolptr, const uint32_t *trowval, const float *tnzval,
                                       const int64_t *ks, bool on_device, PsellDevCSR &C);
// POLEE_DEVICE_BUILD=0 turns the device builder off; so does any of the host builder's experiment knobs
bool psell_device_enabled();
polee_status psell_device_build(polee_ctx *ctx, const PsellDevIn &X, PsellHost &out, PsellDevOut &D, bool want_debug, bool &needs_host);

}  // namespace polee
