"""ctypes binding of libpolee_hip.so (include/polee_hip.h).

The library is built in-tree by `__graft_entry__.build()` (or `make -C polee_amd/csrc`).
There is no CPU fallback: if the shared library is missing, importing the kernels fails
loudly, and without a GPU `Context()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (POLEE_HIP_LIB: another build of the same library -- the toolchain gate test loads csrc/libpolee_hip_untuned.so)
LIB_PATH = os.environ.get("POLEE_HIP_LIB") or os.path.join(_HERE, "csrc", "libpolee_hip.so")

_lib = None

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class PoleeError(RuntimeError):
    """A libpolee_hip call failed; mirrors the reference's error()/@assert exceptions."""

    def __init__(self, status, message):
        super().__init__("polee_hip status %d: %s" % (status, message))
        self.status = status


class NonFiniteError(PoleeError, FloatingPointError):
    """POLEE_ERR_NONFINITE: the @assert isfinite(...) of the reference."""


class ViOpts(C.Structure):
    _fields_ = [("num_steps", C.c_int32), ("num_mc_samples", C.c_int32), ("use_efflen_jacobian", C.c_int32),
                ("gradonly", C.c_int32), ("seed", C.c_uint64), ("z0", f32p), ("y_eps", C.c_double),
                ("adam_initial_learning_rate", C.c_double), ("adam_learning_rate_decay", C.c_double),
                ("adam_min_learning_rate", C.c_double), ("adam_eps", C.c_double), ("adam_rv", C.c_double),
                ("adam_rm", C.c_double), ("max_mu_step", C.c_double), ("max_omega_step", C.c_double),
                ("max_alpha_step", C.c_double), ("profile", C.c_int32), ("deterministic", C.c_int32),
                ("gene_of", C.POINTER(C.c_int32))]


class ViStats(C.Structure):
    _fields_ = [("steps_done", C.c_int32), ("nonfinite_step", C.c_int32), ("loglik_kernel_ms_avg", C.c_double),
                ("loglik_kernel_launches", C.c_int64), ("last_elbo", C.c_double), ("last_lp_mean", C.c_double), ("loglik_pass_ms_avg", C.c_double)]


class LoglikInfo(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("nnz", C.c_int64), ("num_slices", C.c_int64),
                ("num_tiles", C.c_int64), ("padded_nnz", C.c_int64), ("device_bytes", C.c_int64),
                ("stream_bytes", C.c_int64), ("num_empty_rows", C.c_int64), ("max_row_nnz", C.c_int32),
                ("max_tile_cols", C.c_int32), ("stream_rows", C.c_int64 * 8), ("stream_nnz", C.c_int64 * 8),
                ("stream_tiles", C.c_int64 * 8), ("stream_bytes_hbm", C.c_int64 * 8), ("dict_entries", C.c_int64)]


class PsellView(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("nnz", C.c_int64), ("num_slices", C.c_int64),
                ("num_tiles", C.c_int64), ("padded_nnz", C.c_int64), ("num_empty_rows", C.c_int64),
                ("data_bytes", C.c_int64), ("dict_len", C.c_int64), ("max_row_nnz", C.c_int32),
                ("max_tile_cols", C.c_int32), ("data", u8p), ("slice_off", u32p), ("tile_slice", u32p),
                ("tile_dict", u32p), ("dict", u32p), ("row_order", u32p), ("slice_ks", f32p), ("slice_flags", u8p),
                ("num_tiles_a", C.c_int64), ("num_tiles_a1", C.c_int64), ("num_tiles_a1m", C.c_int64), ("num_tiles_a2", C.c_int64), ("slice_w", u8p), ("num_tiles_s", C.c_int64),
                ("stream_rows", C.c_int64 * 8), ("stream_nnz", C.c_int64 * 8), ("stream_bytes", C.c_int64 * 8),
                ("csr_num_rows", C.c_int64), ("csr_rowptr", u32p), ("csr_col", u32p), ("csr_val", f32p), ("csr_rows", u32p),
                ("single_num_rows", C.c_int64), ("single_rows", u32p), ("single_cnt", f32p), ("single_logsum", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libpolee_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C polee_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.polee_last_error.restype = C.c_char_p
        L.polee_last_error.argtypes = [C.c_void_p]
        L.polee_version.restype = C.c_char_p
        L.polee_ctx_stream.restype = C.c_void_p
        L.polee_ctx_stream.argtypes = [C.c_void_p]
        L.polee_ptt_n.restype = C.c_int32
        L.polee_ptt_n.argtypes = [C.c_void_p]
        for name in ("polee_ctx_destroy", "polee_ptt_destroy", "polee_loglik_destroy", "polee_vi_destroy",
                     "polee_approx_destroy", "polee_debug_psell_free", "polee_comm_destroy", "polee_devx_destroy"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        L.polee_vi_default_opts.restype = None
        _lib = L
    return _lib


def check(status, ctx=None):
    if status == 0:
        return
    msg = lib().polee_last_error(ctx).decode("utf-8", "replace")
    if status == 4:
        raise NonFiniteError(status, msg)
    raise PoleeError(status, msg)


def ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
