"""ctypes wrapper of oracle/xbuild_oracle.c -- TEST INFRASTRUCTURE ONLY (parity unpinned: see that file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libxbuild_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "xbuild_oracle.c")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_LIB_PATH)
        _lib.xb_oracle_effective_length.restype = C.c_float
        _lib.xb_oracle_condfragprob.restype = C.c_float
    return _lib


def build(T, F, M, n):
    """T, F, M: the ctypes structs of polee_amd.xbuild.pack (same field layout as the oracle's).  -> the same dict as
    polee_amd.xbuild.build_likelihood_matrix."""
    L = lib()
    eff = np.empty(n, np.float32)
    rows = C.c_int64()
    ptr, cols, vals, rf = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)(), C.POINTER(C.c_int64)()
    L.xb_oracle_build(C.byref(T), C.byref(F), C.byref(M), eff.ctypes.data_as(C.c_void_p), C.byref(rows), C.byref(ptr),
                      C.byref(cols), C.byref(vals), C.byref(rf))
    r = rows.value
    tcolptr = np.ctypeslib.as_array(ptr, shape=(r + 1,)).copy()
    nnz = int(tcolptr[-1] - 1)
    out = dict(m=r, n=n, nnz=nnz, tcolptr=tcolptr,
               trowval=np.ctypeslib.as_array(cols, shape=(max(nnz, 1),))[:nnz].copy(),
               tnzval=np.ctypeslib.as_array(vals, shape=(max(nnz, 1),))[:nnz].copy(), effective_lengths=eff,
               row_fragment=np.ctypeslib.as_array(rf, shape=(max(r, 1),))[:r].copy())
    for q in (ptr, cols, vals, rf):
        L.xb_oracle_free(q)
    return out


def build_biased(T, F, M, B, n, total_bases):
    """BiasedFragModel (second half of xbuild_oracle.c): B = the ctypes struct of polee_amd.xbuild.pack_bias.  -> the dict of
    `build` plus left_bias / right_bias f32 [total_bases]."""
    L = lib()
    eff = np.empty(n, np.float32)
    left, right = np.empty(max(total_bases, 1), np.float32), np.empty(max(total_bases, 1), np.float32)
    rows = C.c_int64()
    ptr, cols, vals, rf = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)(), C.POINTER(C.c_int64)()
    L.xb_oracle_build_biased(C.byref(T), C.byref(F), C.byref(M), C.byref(B), eff.ctypes.data_as(C.c_void_p),
                             left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), C.byref(rows), C.byref(ptr),
                             C.byref(cols), C.byref(vals), C.byref(rf))
    r = rows.value
    tcolptr = np.ctypeslib.as_array(ptr, shape=(r + 1,)).copy()
    nnz = int(tcolptr[-1] - 1)
    out = dict(m=r, n=n, nnz=nnz, tcolptr=tcolptr,
               trowval=np.ctypeslib.as_array(cols, shape=(max(nnz, 1),))[:nnz].copy(),
               tnzval=np.ctypeslib.as_array(vals, shape=(max(nnz, 1),))[:nnz].copy(), effective_lengths=eff,
               row_fragment=np.ctypeslib.as_array(rf, shape=(max(r, 1),))[:r].copy(),
               left_bias=left[:total_bases], right_bias=right[:total_bases])
    for q in (ptr, cols, vals, rf):
        L.xb_oracle_free(q)
    return out
