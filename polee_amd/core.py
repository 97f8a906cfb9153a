"""Host-side mirror of the reference interface over the C ABI (see package docstring)."""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from ._lib import arr, check, ptr, f32p, f64p, i32p, i64p, u32p, u64p

LIKAP_NUM_STEPS = 500      # src/constants.jl:64
LIKAP_NUM_MC_SAMPLES = 6   # src/constants.jl:65


class Context:
    """One GPU + one HIP stream (polee_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(L.lib().polee_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            L.lib().polee_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def mem_info(self):
        """(free, total) device memory in bytes."""
        f, t = C.c_int64(), C.c_int64()
        check(L.lib().polee_ctx_mem_info(self._h, C.byref(f), C.byref(t)), self._h)
        return f.value, t.value

    def synchronize(self):
        check(L.lib().polee_ctx_synchronize(self._h), self._h)

    @property
    def stream(self):
        return L.lib().polee_ctx_stream(self._h)

    def timer_start(self):
        check(L.lib().polee_ctx_timer_start(self._h), self._h)

    def timer_stop(self):
        ms = C.c_double()
        check(L.lib().polee_ctx_timer_stop(self._h, C.byref(ms)), self._h)
        return ms.value


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def version():
    """polee_version(): the library's version, the compiler it was built with and loglik.hip's tuning flags in use."""
    return L.lib().polee_version().decode()


def make_inverse_ptt_params(node_parent_idxs, node_js):
    """src/ptt.jl:293-309 -> (left_index, right_index, leaf_index), 0-based, -1 = none."""
    p, j = arr(node_parent_idxs, np.int32), arr(node_js, np.int32)
    N = p.size
    l, r, f = (np.empty(N, np.int32) for _ in range(3))
    check(L.lib().polee_make_inverse_ptt_params(ptr(p, i32p), ptr(j, i32p), N, ptr(l, i32p), ptr(r, i32p),
                                                ptr(f, i32p)))
    return l, r, f


class PolyaTreeTransform:
    """src/ptt.jl:6-27.  Built from the serialised tree (ptt.jl:89-116) or from the TF-op
    index arrays (left, right, leaf)."""

    def __init__(self, node_parent_idxs=None, node_js=None, ctx=None, index=None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        if index is not None:
            l, r, f = (arr(a, np.int32).reshape(-1) for a in index)
            check(L.lib().polee_ptt_create_from_index(self.ctx._h, ptr(l, i32p), ptr(r, i32p), ptr(f, i32p),
                                                      int(l.size), C.byref(self._h)), self.ctx._h)
            self.node_parent_idxs = self.node_js = None
        else:
            self.node_parent_idxs = arr(node_parent_idxs, np.int32)
            self.node_js = arr(node_js, np.int32)
            if self.node_parent_idxs.size != self.node_js.size:
                raise ValueError("node_parent_idxs and node_js differ in length")  # @assert ptt.jl:91
            check(L.lib().polee_ptt_create(self.ctx._h, ptr(self.node_parent_idxs, i32p), ptr(self.node_js, i32p),
                                           int(self.node_js.size), C.byref(self._h)), self.ctx._h)
        self.n = int(L.lib().polee_ptt_n(self._h))

    def __del__(self):
        try:
            if self._h:
                L.lib().polee_ptt_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _rows(self, a, dtype, width):
        a = arr(a, dtype)
        single = a.ndim == 1
        a = a.reshape(1, -1) if single else a
        if a.shape[1] != width:
            raise ValueError("expected rows of length %d, got %d" % (width, a.shape[1]))
        return a, single

    def transform(self, ys, compute_ladj=False):
        """transform!(t, ys, xs, Val(compute_ladj)) (ptt.jl:125-160) -> (xs, ladj)."""
        ys, single = self._rows(ys, np.float64, self.n - 1)
        B = ys.shape[0]
        xs = np.empty((B, self.n), np.float32)
        ladj = np.zeros(B, np.float64) if compute_ladj else None
        check(L.lib().polee_ptt_transform(self._h, ptr(ys, f64p), B, ptr(xs, f32p), ptr(ladj, f64p)), self.ctx._h)
        if single:
            return xs[0], (float(ladj[0]) if compute_ladj else 0.0)
        return xs, (ladj if compute_ladj else np.zeros(B))

    def transform_gradients(self, ys, x_grad, with_ladj=True):
        """transform_gradients! (ptt.jl:167-209) / transform_gradients_no_ladj! (:217-251) -> y_grad (f64)."""
        ys, single = self._rows(ys, np.float64, self.n - 1)
        xg, _ = self._rows(x_grad, np.float64, self.n)
        B = ys.shape[0]
        yg = np.empty((B, self.n - 1), np.float64)
        check(L.lib().polee_ptt_transform_gradients(self._h, ptr(ys, f64p), ptr(xg, f64p), B, int(with_ladj),
                                                    ptr(yg, f64p)), self.ctx._h)
        return yg[0] if single else yg

    def transform_gradients_no_ladj(self, ys, x_grad):
        return self.transform_gradients(ys, x_grad, with_ladj=False)

    def inverse_transform(self, xs):
        """inverse_transform! (ptt.jl:257-285) -> (ys, ladj)."""
        xs, single = self._rows(xs, np.float32, self.n)
        B = xs.shape[0]
        ys = np.empty((B, self.n - 1), np.float64)
        ladj = np.zeros(B, np.float64)
        check(L.lib().polee_ptt_inverse_transform(self._h, ptr(xs, f32p), B, ptr(ys, f64p), ptr(ladj, f64p)),
              self.ctx._h)
        return (ys[0], float(ladj[0])) if single else (ys, ladj)


def _tree_for_ops(left_index, right_index, leaf_index, ctx):
    l = arr(left_index, np.int32)
    if l.ndim == 2 and l.shape[0] > 1:
        raise ValueError("per-row trees: build one PolyaTreeTransform per tree (or use RNASeqApproxLikelihood)")
    return PolyaTreeTransform(ctx=ctx, index=(l, right_index, leaf_index))


def hsb(y_logit, left_index, right_index=None, leaf_index=None, ctx=None):
    """TF op HSB (hsb_ops.cpp:17-120)."""
    t = left_index if isinstance(left_index, PolyaTreeTransform) else _tree_for_ops(left_index, right_index, leaf_index, ctx)
    y, single = t._rows(y_logit, np.float32, t.n - 1)
    x = np.empty((y.shape[0], t.n), np.float32)
    check(L.lib().polee_hsb(t._h, ptr(y, f32p), y.shape[0], ptr(x, f32p)), t.ctx._h)
    return x


def inv_hsb(x, left_index, right_index=None, leaf_index=None, ctx=None):
    """TF op InvHSB (hsb_ops.cpp:128-249) -> (y f64 [B,n-1], ladj f32 [B,1])."""
    t = left_index if isinstance(left_index, PolyaTreeTransform) else _tree_for_ops(left_index, right_index, leaf_index, ctx)
    x, _ = t._rows(x, np.float32, t.n)
    B = x.shape[0]
    y = np.empty((B, t.n - 1), np.float64)
    ladj = np.empty((B, 1), np.float32)
    check(L.lib().polee_inv_hsb(t._h, ptr(x, f32p), B, ptr(y, f64p), ptr(ladj, f32p)), t.ctx._h)
    return y, ladj


def inv_hsb_grad(y_grad, ladj_grad, y, left_index, right_index=None, leaf_index=None, ctx=None):
    """TF op InvHSBGrad (hsb_ops.cpp:252-402) -> backprops f32 [B,n]."""
    t = left_index if isinstance(left_index, PolyaTreeTransform) else _tree_for_ops(left_index, right_index, leaf_index, ctx)
    yg, _ = t._rows(y_grad, np.float64, t.n - 1)
    yy, _ = t._rows(y, np.float64, t.n - 1)
    B = yy.shape[0]
    lg = arr(np.reshape(ladj_grad, (-1,)), np.float32)
    bp = np.empty((B, t.n), np.float32)
    check(L.lib().polee_inv_hsb_grad(t._h, ptr(yg, f64p), ptr(lg, f32p), ptr(yy, f64p), B, ptr(bp, f32p)), t.ctx._h)
    return bp


class DeviceX:
    """X by columns (1-based CSC, the likelihood-matrix HDF5 arrays) in device memory, uploaded once for the two device builders
    that read it (polee_devx_upload): `RNASeqSample(..., devx=dx)` and `hclust(..., devx=dx)`.  Read-only: both may run at once,
    each on its own context of the same device."""

    def __init__(self, m, n, colptr, rowval, nzval, ctx=None):
        self.ctx = ctx or default_context()
        self.m, self.n = int(m), int(n)
        colptr = np.ascontiguousarray(colptr)
        if colptr.dtype not in (np.dtype(np.uint32), np.dtype(np.uint64)):
            colptr = colptr.astype(np.uint64)
        rowval = arr(rowval, np.uint32)
        nzval = None if nzval is None else arr(nzval, np.float32)  # (None: upload_values follows)
        self._csc = (colptr, rowval)
        self._h = C.c_void_p()
        check(L.lib().polee_devx_upload(self.ctx._h, C.c_int64(self.m), C.c_int64(self.n), colptr.ctypes.data_as(C.c_void_p),
                                        int(colptr.dtype.itemsize), ptr(rowval, u32p), ptr(nzval, f32p), C.byref(self._h)), self.ctx._h)

    def upload_values(self, nzval):
        """the non-zeros' values after the fact: the tree reads colptr + rowval only and may already be running"""
        nzval = arr(nzval, np.float32)
        check(L.lib().polee_devx_upload_values(self._h, ptr(nzval, f32p)), self.ctx._h)

    def __del__(self):
        try:
            if self._h:
                L.lib().polee_devx_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class RNASeqSample:
    """The numeric part of RNASeqSample (src/rnaseq_sample.jl:6-23): X (m x n CSC, 1-based
    colptr/rowval exactly as in the likelihood-matrix HDF5, :505-519) + effective_lengths,
    resident on the GPU.  `ks` = row multiplicities for the factored likelihood."""

    def __init__(self, m, n, colptr, rowval, nzval, effective_lengths=None, ks=None, ctx=None, xt=None, _xbuild=None, devx=None):
        self.ctx = ctx or default_context()
        self.m, self.n = int(m), int(n)
        self.effective_lengths = None if effective_lengths is None else arr(effective_lengths, np.float32)
        self._h = C.c_void_p()
        self._csc = None
        ks_a = None if ks is None else arr(ks, np.int64)
        if _xbuild is not None:  # an xbuild result, still on the device (polee_amd.xbuild.build_likelihood_matrix(return_sample=True))
            check(L.lib().polee_loglik_create_from_xbuild(self.ctx._h, _xbuild, ptr(ks_a, i64p), C.byref(self._h)), self.ctx._h)
        elif devx is not None:  # X already on the device (DeviceX): no second upload (nzval: only if the handle has no values yet)
            self._csc = devx._csc
            nz = None if nzval is None else arr(nzval, np.float32)
            check(L.lib().polee_loglik_create_from_devx(self.ctx._h, devx._h, ptr(nz, f32p), ptr(ks_a, i64p), C.byref(self._h)), self.ctx._h)
        elif xt is not None:
            tp, tr, tv = arr(xt[0], np.uint64), arr(xt[1], np.uint32), arr(xt[2], np.float32)
            check(L.lib().polee_loglik_create_from_xt(self.ctx._h, C.c_int64(self.m), C.c_int64(self.n), ptr(tp, u64p),
                                                      ptr(tr, u32p), ptr(tv, f32p), ptr(ks_a, i64p),
                                                      C.byref(self._h)), self.ctx._h)
        else:
            colptr = np.ascontiguousarray(colptr)
            if colptr.dtype not in (np.dtype(np.uint32), np.dtype(np.uint64)):
                colptr = colptr.astype(np.uint64)
            rowval, nzval = arr(rowval, np.uint32), arr(nzval, np.float32)
            self._csc = (colptr, rowval)  # kept for tree construction (hclust)
            check(L.lib().polee_loglik_create(self.ctx._h, C.c_int64(self.m), C.c_int64(self.n),
                                              colptr.ctypes.data_as(C.c_void_p), int(colptr.dtype.itemsize),
                                              ptr(rowval, u32p), ptr(nzval, f32p), ptr(ks_a, i64p),
                                              C.byref(self._h)), self.ctx._h)
        self.has_ks = ks is not None

    def __del__(self):
        try:
            if self._h:
                L.lib().polee_loglik_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def built_on_device(self):
        """True when the device layout was built by the device builder (csrc/psell_device.hip), False: by the host builder."""
        return bool(L.lib().polee_loglik_built_on_device(self._h))

    def set_deterministic(self, on=True):
        """Fixed-order (bitwise reproducible) gradient sums instead of float atomics (polee_loglik_set_deterministic)."""
        check(L.lib().polee_loglik_set_deterministic(self._h, int(bool(on))), self.ctx._h)

    @property
    def info(self):
        i = L.LoglikInfo()
        check(L.lib().polee_loglik_get_info(self._h, C.byref(i)))
        return {k: (list(getattr(i, k)) if hasattr(getattr(i, k), "__len__") else getattr(i, k)) for k, _ in i._fields_}

    def log_likelihood(self, xs, gradonly=False):
        """log_likelihood (likelihood.jl:36-56) for one vector [n] or K stacked vectors [K, n]
        -> (lp, x_grad); lp is 0.0 when gradonly."""
        xs = arr(xs, np.float32)
        single = xs.ndim == 1
        xs2 = xs.reshape(1, -1) if single else xs
        K = xs2.shape[0]
        if xs2.shape[1] != self.n:
            raise ValueError("xs must have %d columns" % self.n)
        g = np.empty((K, self.n), np.float64)
        lp = None if gradonly else np.zeros(K, np.float64)
        check(L.lib().polee_loglik_eval(self._h, ptr(xs2, f32p), K, ptr(g, f64p), ptr(lp, f64p)), self.ctx._h)
        if gradonly:
            lp = np.zeros(K)
        return (float(lp[0]), g[0]) if single else (lp, g)


def log_likelihood(sample, xs, gradonly=False):
    return sample.log_likelihood(xs, gradonly)


def factored_log_likelihood(sample, xs, gradonly=False):
    """factored_log_likelihood (likelihood.jl:59-85); the sample must have been built with ks."""
    if not sample.has_ks:
        raise ValueError("sample was created without multiplicities ks")
    return sample.log_likelihood(xs, gradonly)


def effective_length_jacobian_adjustment(efflens, xs, x_grad, ctx=None):
    """effective_length_jacobian_adjustment! (likelihood.jl:93-110) -> (xls, adjusted x_grad)."""
    ctx = ctx or default_context()
    efflens, xs = arr(efflens, np.float32), arr(xs, np.float32)
    single = xs.ndim == 1
    xs2 = xs.reshape(1, -1) if single else xs
    K, n = xs2.shape
    g = arr(x_grad, np.float64).reshape(K, n).copy()
    xls = np.empty((K, n), np.float32)
    check(L.lib().polee_efflen_jacobian_adjustment(ctx._h, ptr(efflens, f32p), ptr(xs2, f32p), K, C.c_int64(n),
                                                   ptr(g, f64p), ptr(xls, f32p)), ctx._h)
    return (xls[0], g[0]) if single else (xls, g)


def _gene_of(gene_transcripts, n):
    """The reference's Dict{gene id -> 1-based transcript indexes} (likelihood-approximation.jl:476-487), or an int array
    gene_of[n] (0-based gene index, -1 = no gene known), as the int32 array the C ABI takes."""
    if isinstance(gene_transcripts, dict):
        gene_of = np.full(n, -1, np.int32)
        for gi, idxs in enumerate(gene_transcripts.values()):
            for i in idxs:
                if not 1 <= i <= n:
                    raise ValueError("transcript index %d out of range" % i)
                gene_of[i - 1] = gi
        return gene_of
    gene_of = arr(gene_transcripts, np.int32).reshape(-1)
    if gene_of.size != n:
        raise ValueError("gene_of must have one entry per transcript")
    return gene_of


def gene_noninformative_prior(efflens, xls, xs, x_grad, gene_transcripts, ctx=None):
    """gene_noninformative_prior! (likelihood.jl:114-159) -> adjusted x_grad.

    `gene_transcripts` is the reference's Dict{gene id -> 1-based transcript indexes}
    (likelihood-approximation.jl:476-487) or an int array gene_of[n] (0-based gene index, -1 = none)."""
    ctx = ctx or default_context()
    efflens, xs, xls = arr(efflens, np.float32), arr(xs, np.float32), arr(xls, np.float32)
    single = xs.ndim == 1
    xs2 = xs.reshape(1, -1) if single else xs
    K, n = xs2.shape
    gene_of = _gene_of(gene_transcripts, n)
    g = arr(x_grad, np.float64).reshape(K, n).copy()
    check(L.lib().polee_gene_noninformative_prior(ctx._h, ptr(efflens, f32p), ptr(xls.reshape(K, n), f32p),
                                                  ptr(xs2, f32p), K, C.c_int64(n), ptr(gene_of, L.i32p), ptr(g, f64p)),
          ctx._h)
    return g[0] if single else g


def _vecs(dtype, *arrs):
    out = [arr(a, dtype).reshape(-1) for a in arrs]
    if len({a.size for a in out}) != 1:
        raise ValueError("arguments differ in length")
    return out


def logit_normal_transform(mu, sigma, zs, compute_ladj=False, ctx=None):
    """logit_normal_transform! (logitnormal.jl:8-20) -> (ys f64, ladj)."""
    ctx = ctx or default_context()
    mu, sigma, zs = _vecs(np.float32, mu, sigma, zs)
    ys = np.empty(mu.size, np.float64)
    ladj = C.c_double(0.0)
    check(L.lib().polee_logit_normal_transform(ctx._h, ptr(mu, f32p), ptr(sigma, f32p), ptr(zs, f32p),
                                               C.c_int64(mu.size), ptr(ys, f64p),
                                               C.byref(ladj) if compute_ladj else None), ctx._h)
    return ys, ladj.value


def logit_normal_transform_gradients(zs, ys, mu, sigma, y_grad, z_grad=None, mu_grad=None, sigma_grad=None, ctx=None):
    """logit_normal_transform_gradients! (logitnormal.jl:23-55): returns (z_grad, mu_grad, sigma_grad), each the
    given array (or zeros) plus this call's contribution."""
    ctx = ctx or default_context()
    zs, sigma, y_grad = _vecs(np.float32, zs, sigma, y_grad)
    ys = arr(ys, np.float64).reshape(-1)
    n = zs.size
    zg = np.zeros(n, np.float32) if z_grad is None else arr(z_grad, np.float32).copy()
    mg = np.zeros(n, np.float32) if mu_grad is None else arr(mu_grad, np.float32).copy()
    sg = np.zeros(n, np.float32) if sigma_grad is None else arr(sigma_grad, np.float32).copy()
    check(L.lib().polee_logit_normal_transform_gradients(ctx._h, ptr(zs, f32p), ptr(ys, f64p), ptr(sigma, f32p),
                                                         ptr(y_grad, f32p), C.c_int64(n), ptr(zg, f32p),
                                                         ptr(mg, f32p), ptr(sg, f32p)), ctx._h)
    return zg, mg, sg


def sinh_asinh_transform(alpha, zs0, compute_ladj=False, ctx=None):
    """sinh_asinh_transform! (sinh_arcsinh.jl:10-23) -> (zs, ladj)."""
    ctx = ctx or default_context()
    alpha, zs0 = _vecs(np.float32, alpha, zs0)
    zs = np.empty(alpha.size, np.float32)
    ladj = C.c_double(0.0)
    check(L.lib().polee_sinh_asinh_transform(ctx._h, ptr(alpha, f32p), ptr(zs0, f32p), C.c_int64(alpha.size),
                                             ptr(zs, f32p), C.byref(ladj) if compute_ladj else None), ctx._h)
    return zs, ladj.value


def sinh_asinh_transform_gradients(zs0, alpha, z_grad, alpha_grad=None, ctx=None):
    """sinh_asinh_transform_gradients! (sinh_arcsinh.jl:29-38)."""
    ctx = ctx or default_context()
    zs0, alpha, z_grad = _vecs(np.float32, zs0, alpha, z_grad)
    ag = np.zeros(zs0.size, np.float32) if alpha_grad is None else arr(alpha_grad, np.float32).copy()
    check(L.lib().polee_sinh_asinh_transform_gradients(ctx._h, ptr(zs0, f32p), ptr(alpha, f32p), ptr(z_grad, f32p),
                                                       C.c_int64(zs0.size), ptr(ag, f32p)), ctx._h)
    return ag


def kumaraswamy_transform(as_, bs, zs, compute_ladj=True, ctx=None):
    """kumaraswamy_transform! (kumaraswamy.jl:27-51) -> (ys, ladj)."""
    ctx = ctx or default_context()
    as_, bs, zs = _vecs(np.float32, as_, bs, zs)
    ys = np.empty(zs.size, np.float64)
    ladj = C.c_double(0.0)
    check(L.lib().polee_kumaraswamy_transform(ctx._h, ptr(as_, f32p), ptr(bs, f32p), ptr(zs, f32p), C.c_int64(zs.size),
                                              ptr(ys, f64p), C.byref(ladj) if compute_ladj else None), ctx._h)
    return ys, ladj.value


def kumaraswamy_transform_gradients(zs, as_, bs, y_grad, a_grad=None, b_grad=None, ctx=None):
    """kumaraswamy_transform_gradients! (kumaraswamy.jl:54-78)."""
    ctx = ctx or default_context()
    zs, as_, bs, y_grad = _vecs(np.float32, zs, as_, bs, y_grad)
    ag = np.zeros(zs.size, np.float32) if a_grad is None else arr(a_grad, np.float32).copy()
    bg = np.zeros(zs.size, np.float32) if b_grad is None else arr(b_grad, np.float32).copy()
    check(L.lib().polee_kumaraswamy_transform_gradients(ctx._h, ptr(zs, f32p), ptr(as_, f32p), ptr(bs, f32p),
                                                        ptr(y_grad, f32p), C.c_int64(zs.size), ptr(ag, f32p),
                                                        ptr(bg, f32p)), ctx._h)
    return ag, bg


def host_cache_trim():
    """Release the scratch blocks the host-side builders keep between samples (polee_host_cache_trim)."""
    f = L.lib().polee_host_cache_trim
    f.restype, f.argtypes = None, []
    f()


def host_cache_configure(cap_mb=-1):
    """Cap (MB) of the builders' scratch-block cache; frees what no longer fits.  cap_mb < 0: query only.  Returns the cap
    in force.  (Default: a quarter of the memory available to the process, at most 8 GiB -- a process keeps up to that
    much resident between samples.)"""
    f = L.lib().polee_host_cache_configure
    f.restype, f.argtypes = C.c_int64, [C.c_int64]
    return int(f(int(cap_mb)))


def host_cache_bytes():
    f = L.lib().polee_host_cache_bytes
    f.restype, f.argtypes = C.c_int64, []
    return int(f())


def device_cache_poison_stats():
    """Debug mode POLEE_DEVICE_CACHE_POISON=1 (csrc/common.hpp): (blocks verified, blocks found overwritten after their release,
    words overwritten) so far; zeros when the mode is off."""
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    L.lib().polee_debug_device_cache_poison(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def device_cache_bytes():
    """Device bytes the library keeps for its next allocations (polee_device_cache_bytes; host_cache_trim() frees them)."""
    f = L.lib().polee_device_cache_bytes
    f.restype, f.argtypes = C.c_int64, []
    return int(f())


def hclust(m, n, colptr, rowval, parallel=False, device=False, ctx=None, devx=None):
    """hclust + order_nodes (hclust.jl:193-319, 361-389): the tree heuristic behind PolyaTreeTransform(X, :cluster)
    (ptt.jl:35-52).  X in CSC, 1-based (likelihood-matrix HDF5 arrays) -> (node_parent_idxs, node_js), int32 [2n-1],
    i.e. what the prep HDF5 stores and PolyaTreeTransform(...) takes.  Runs on the host, as in the reference.
    parallel=True: the same joining rule in rounds of mutually-best merges on all host threads (polee_hclust_parallel;
    a documented variant, not the reference's tree node for node).  device=True: that variant on the GPU
    (polee_hclust_parallel_device: the same arrays as parallel=True); devx: from a DeviceX instead of the host arrays."""
    if devx is not None:
        ctx = ctx or devx.ctx
        parents, js = np.empty(2 * int(devx.n) - 1, np.int32), np.empty(2 * int(devx.n) - 1, np.int32)
        check(L.lib().polee_hclust_parallel_device_from_devx(ctx._h, devx._h, ptr(parents, L.i32p), ptr(js, L.i32p)), ctx._h)
        return parents, js
    colptr = np.ascontiguousarray(colptr)
    if colptr.dtype not in (np.dtype(np.uint32), np.dtype(np.uint64)):
        colptr = colptr.astype(np.uint64)
    rowval = arr(rowval, np.uint32)
    parents, js = np.empty(2 * int(n) - 1, np.int32), np.empty(2 * int(n) - 1, np.int32)
    if device:
        ctx = ctx or default_context()
        check(L.lib().polee_hclust_parallel_device(ctx._h, C.c_int64(int(m)), C.c_int64(int(n)), colptr.ctypes.data_as(C.c_void_p),
                                                   int(colptr.dtype.itemsize), ptr(rowval, u32p), ptr(parents, L.i32p), ptr(js, L.i32p)), ctx._h)
        return parents, js
    f = L.lib().polee_hclust_parallel if parallel else L.lib().polee_hclust
    check(f(C.c_int64(int(m)), C.c_int64(int(n)), colptr.ctypes.data_as(C.c_void_p), int(colptr.dtype.itemsize),
            ptr(rowval, u32p), ptr(parents, L.i32p), ptr(js, L.i32p)))
    return parents, js


class LogitSkewNormalPTTApprox:
    """src/likelihood-approximation.jl:8-16.  treemethod "cluster" (hclust.jl) or "sequential" (list tree) is used
    by approximate_likelihood when no tree is passed (ptt.jl:35-52)."""

    def __init__(self, treemethod="cluster"):
        self.treemethod = treemethod


class Comm:
    """RCCL communicator over the GPUs that share ONE sample by rows (polee_comm).

    `broadcast` distributes rank 0's 128-byte id: a callable bytes -> bytes (e.g. built on
    torch.distributed.broadcast_object_list or MPI); it may be None for a single rank."""

    def __init__(self, ctx, world_size=1, rank=0, broadcast=None):
        self.ctx, self.world_size, self.rank = ctx, int(world_size), int(rank)
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            check(L.lib().polee_comm_unique_id(uid))
        raw = bytes(uid)
        if self.world_size > 1:
            if broadcast is None:
                raise ValueError("a broadcast function is required for more than one rank")
            raw = broadcast(raw)
        uid = (C.c_uint8 * 128).from_buffer_copy(raw)
        self._h = C.c_void_p()
        check(L.lib().polee_comm_create(ctx._h, self.world_size, self.rank, uid, C.byref(self._h)), ctx._h)

    def __del__(self):
        try:
            if self._h:
                L.lib().polee_comm_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def info(self):
        """What the transport reports (polee_comm_info): dict(transport "rccl" | "host", count, rank) -- for RCCL the
        communicator's own ncclCommCount / ncclCommUserRank."""
        t, n, r = C.c_int32(), C.c_int32(), C.c_int32()
        check(L.lib().polee_comm_info(self._h, C.byref(t), C.byref(n), C.byref(r)), self.ctx._h)
        import os
        return dict(transport={1: "rccl", 2: "host"}.get(t.value, "?"), count=n.value, rank=r.value,
                    # (RCCL transport: POLEE_COMM_ALGO=rs_ag exchanges by reduce-scatter + all-gather instead of one all-reduce)
                    algo=os.environ.get("POLEE_COMM_ALGO", "allreduce") if t.value == 1 else "host all-reduce")

    def allreduce_ms(self, count, reps=20):
        """HIP-event time of one all-reduce of `count` f32 on the library's stream, averaged over `reps` calls behind an untimed
        first one (polee_debug_comm_allreduce_ms): what the exchange itself adds to a row-sharded pass or a regression step."""
        ms = C.c_double()
        check(L.lib().polee_debug_comm_allreduce_ms(self._h, C.c_int64(int(count)), C.c_int32(int(reps)), C.byref(ms)), self.ctx._h)
        return ms.value

    def allreduce_sum(self, values):
        """Sum of a float32 array over the ranks (polee_allreduce_sum_f32)."""
        v = arr(values, np.float32).copy()
        check(L.lib().polee_allreduce_sum_f32(self._h, ptr(v, f32p), C.c_int64(v.size)), self.ctx._h)
        return v


class HostComm(Comm):
    """The communicator interface over a HOST all-reduce supplied by the caller (polee_comm_create_host): `allreduce`
    receives a float32 / float64 NumPy array and must sum it over all ranks IN PLACE (MPI Allreduce, a torch.distributed
    gloo group, ...).  For clusters without RCCL between the ranks and for tests with several ranks on one GPU."""

    _CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

    def __init__(self, ctx, world_size, rank, allreduce):
        self.ctx, self.world_size, self.rank = ctx, int(world_size), int(rank)

        def cb(_user, buf, count, is_f64):
            try:
                ct = C.c_double if is_f64 else C.c_float
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ct)), shape=(int(count),))
                allreduce(a)
                return 0
            except Exception:  # (an exception must not unwind through the C frames)
                import traceback
                traceback.print_exc()
                return 1
        self._cb = HostComm._CB(cb)  # (kept alive with the handle)
        self._h = C.c_void_p()
        check(L.lib().polee_comm_create_host(ctx._h, self.world_size, self.rank, self._cb, None, C.byref(self._h)), ctx._h)


class LikelihoodApproximationFit:
    """State of one fit (polee_vi): lets callers step the VI loop and inspect it."""

    def __init__(self, sample, t, efflens=None, num_steps=LIKAP_NUM_STEPS, num_mc_samples=LIKAP_NUM_MC_SAMPLES,
                 use_efflen_jacobian=True, gradonly=True, seed=123456789, z0=None, profile=False, comm=None,
                 gene_transcripts=None, adam=None, deterministic=None):
        """gene_transcripts (Dict gene -> 1-based transcript indexes, or gene_of int[n]): gene_noninformative = true.
        deterministic: fixed-order gradient sums (bitwise reproducible) instead of float atomics; None = on when the sample is
        shared by more than one rank (comm.world_size > 1: repeated N-rank fits of one sample are bitwise stable),
        off otherwise.
        adam: optional overrides of the optimiser constants of polee_vi_opts (adam_initial_learning_rate, adam_rm,
        max_mu_step, ...; defaults = the reference's, constants.jl:48-65)."""
        self.sample, self.t, self.ctx = sample, t, sample.ctx
        self.comm = comm
        efflens = sample.effective_lengths if efflens is None else efflens
        if efflens is None:
            raise ValueError("effective lengths are required")
        self.efflens = arr(efflens, np.float32)
        o = L.ViOpts()
        L.lib().polee_vi_default_opts(C.byref(o))
        o.num_steps, o.num_mc_samples = int(num_steps), int(num_mc_samples)
        o.use_efflen_jacobian, o.gradonly, o.seed, o.profile = int(use_efflen_jacobian), int(gradonly), int(seed), int(profile)
        self._z0 = None
        if z0 is not None:
            self._z0 = arr(z0, np.float32).reshape(-1)
            if self._z0.size != num_steps * num_mc_samples * (sample.n - 1):
                raise ValueError("z0 must have num_steps*num_mc_samples*(n-1) elements")
            o.z0 = ptr(self._z0, f32p)
        # (polee_vi_opts.deterministic: 0 = the library's rule -- on exactly when the sample is shared by more than one rank --, 1 on, -1 off)
        o.deterministic = 0 if deterministic is None else (1 if deterministic else -1)
        self.deterministic = (comm is not None and getattr(comm, "world_size", 1) > 1) if deterministic is None else bool(deterministic)
        for key, val in (adam or {}).items():
            if not (key.startswith("adam_") or key.startswith("max_")) or not hasattr(o, key):
                raise ValueError("unknown optimiser constant %r" % (key,))
            setattr(o, key, float(val))
        self._gene_of = None
        if gene_transcripts is not None:
            self._gene_of = _gene_of(gene_transcripts, sample.n)
            if (self._gene_of >= 0).any():
                o.gene_of = ptr(self._gene_of, L.i32p)
            else:  # (likelihood-approximation.jl:487-490)
                import warnings
                warnings.warn("'--gene-noninformative' used, but no gene information available")
        self.opts = o
        self.n, self.K = sample.n, int(num_mc_samples)
        self._h = C.c_void_p()
        check(L.lib().polee_vi_create(sample._h, t._h, ptr(self.efflens, f32p), C.byref(o), C.byref(self._h)),
              self.ctx._h)
        if comm is not None:  # `sample` holds this rank's block of fragments (cohort.shard_rows)
            check(L.lib().polee_vi_set_comm(self._h, comm._h), self.ctx._h)

    def __del__(self):
        try:
            if self._h:
                L.lib().polee_vi_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def run(self, nsteps):
        check(L.lib().polee_vi_run(self._h, int(nsteps)), self.ctx._h)

    def sync(self):
        check(L.lib().polee_vi_sync(self._h), self.ctx._h)

    def params(self):
        nm1 = self.n - 1
        mu, om, al = (np.empty(nm1, np.float32) for _ in range(3))
        check(L.lib().polee_vi_get_params(self._h, ptr(mu, f32p), ptr(om, f32p), ptr(al, f32p)), self.ctx._h)
        return mu, om, al

    def set_params(self, mu=None, omega=None, alpha=None):
        a = [None if v is None else arr(v, np.float32) for v in (mu, omega, alpha)]
        check(L.lib().polee_vi_set_params(self._h, ptr(a[0], f32p), ptr(a[1], f32p), ptr(a[2], f32p)), self.ctx._h)

    def stats(self):
        s = L.ViStats()
        check(L.lib().polee_vi_get_stats(self._h, C.byref(s)), self.ctx._h)
        return {k: getattr(s, k) for k, _ in s._fields_}

    def trace(self):
        n = self.stats()["steps_done"]
        e, l = np.zeros(n), np.zeros(n)
        check(L.lib().polee_vi_get_trace(self._h, ptr(e, f64p), ptr(l, f64p)), self.ctx._h)
        return e, l

    def export_noise(self, step):
        z = np.empty((self.K, self.n - 1), np.float32)
        check(L.lib().polee_vi_export_noise(self._h, int(step), ptr(z, f32p)), self.ctx._h)
        return z

    def eval_gradients(self):
        n, nm1, K = self.n, self.n - 1, self.K
        out = dict(xs=np.empty((K, n), np.float32), x_grad=np.empty((K, n), np.float64),
                   y_grad=np.empty((K, nm1), np.float64), mu_grad=np.empty(nm1, np.float32),
                   omega_grad=np.empty(nm1, np.float32), alpha_grad=np.empty(nm1, np.float32),
                   lp=np.empty(K, np.float64), ladj=np.empty(K, np.float64))
        check(L.lib().polee_vi_eval_gradients(
            self._h, ptr(out["xs"], f32p), ptr(out["x_grad"], f64p), ptr(out["y_grad"], f64p),
            ptr(out["mu_grad"], f32p), ptr(out["omega_grad"], f32p), ptr(out["alpha_grad"], f32p),
            ptr(out["lp"], f64p), ptr(out["ladj"], f64p)), self.ctx._h)
        return out


import threading as _threading
_host_tree_slot = _threading.Semaphore(1)  # treemethod "cluster_auto": one host-built tree at a time (it takes every host thread)
_fits_lock = _threading.Lock()
_fits_in_flight = [0]  # samples of this process being laid out or fitted on the GPU right now ("cluster_auto": with no OTHER one, the GPU is free for the tree)


def sample_and_tree(approx, m, n, colptr, rowval, nzval, effective_lengths, ctx=None, ks=None):
    """The two host-side stages of preparing a sample side by side: the tree heuristic (hclust*) on a helper thread while
    this thread builds the sample's device layout (RNASeqSample); both are C calls that release the GIL, and neither
    needs the other.  Returns (sample, PolyaTreeTransform) -- what `approximate_likelihood(approx, sample, t)` takes.
    (`approximate_likelihood(approx, sample)` alone does the same two stages one after the other.)"""
    from concurrent.futures import ThreadPoolExecutor
    ctx = ctx or default_context()
    tm = approx.treemethod
    with _fits_lock:
        others = _fits_in_flight[0]
        _fits_in_flight[0] += 1
    try:
        return _sample_and_tree(approx, tm, m, n, colptr, rowval, nzval, effective_lengths, ctx, ks, others)
    finally:
        with _fits_lock:
            _fits_in_flight[0] -= 1


def _sample_and_tree(approx, tm, m, n, colptr, rowval, nzval, effective_lengths, ctx, ks, others):
    from concurrent.futures import ThreadPoolExecutor
    #  the tree job runs beside the layout build on a helper thread: it gets a context -- a HIP stream and an error
    # slot -- of its OWN on the same device, kept with the caller's context.  On one shared context the two builders' kernels
    # queued behind each other on one stream, every stream synchronisation of either waited for both, and both threads wrote
    # the context's error string.
    tree_ctx = getattr(ctx, "_tree_ctx", None)
    if tree_ctx is None and tm in ("cluster_auto", "cluster_device"):
        tree_ctx = ctx._tree_ctx = Context(ctx.device)
    # One device copy of X when the tree is built on the device too: 1.9 GB over PCIe at C2 instead of 2.9 GB.
    # POLEE_SHARED_X=0: each builder uploads its own (A/B).
    share_x = os.environ.get("POLEE_SHARED_X", "1") != "0" and int(m) < 2 ** 32 - 1
    with ThreadPoolExecutor(max_workers=1) as pool:
        fut = None
        devx = None

        def shared():
            try:
                return DeviceX(m, n, colptr, rowval, None, ctx=ctx) if share_x else None  # (the values go up with the layout, beside its first kernels)
            except L.PoleeError as e:  # (POLEE_ERR_UNSUPPORTED: more than 32 bits of rows or non-zeros -- the host paths)
                if e.status == 5:
                    return None
                raise
        if tm == "cluster_auto":
            # The rounds variant is the SAME tree from the host (all host threads) and from the GPU: a cohort that is bound by the
            # GPU gives the tree to the host CPUs whenever they are idle -- one host tree at a time -- and to the GPU otherwise.
            # (the host only when the GPU has a fit to run meanwhile -- a lone worker finds it idle and builds the tree there, 0.09 s
            # instead of 0.45 s -- or when the slot is shared with other processes, whose fits this one cannot see)
            busy = others > 0 or not isinstance(_host_tree_slot, _threading.Semaphore)
            on_host = busy and _host_tree_slot.acquire(False)  # (non-blocking; positional: threading and multiprocessing name the argument differently)
            if not on_host:
                devx = shared()

            def tree_job():
                try:
                    return hclust(m, n, colptr, rowval, parallel=on_host, device=not on_host, ctx=tree_ctx, devx=devx)
                finally:
                    if on_host:
                        _host_tree_slot.release()
            fut = pool.submit(tree_job)
        elif tm in ("cluster", "cluster_parallel", "cluster_device"):
            if tm == "cluster_device":
                devx = shared()
            fut = pool.submit(hclust, m, n, colptr, rowval, tm == "cluster_parallel", tm == "cluster_device", tree_ctx or ctx, devx)
        sample = RNASeqSample(m, n, colptr, rowval, nzval, effective_lengths, ks=ks, ctx=ctx, devx=devx)
        if fut is not None:
            parents, js = fut.result()
        elif tm == "sequential":
            parents, js = list_nodes(int(n))
        else:
            raise ValueError("%r is not a supported Polya tree transform heuristic" % (tm,))
    return sample, PolyaTreeTransform(parents, js, ctx=ctx)


def approximate_likelihood(approx, sample, t=None, gene_noninformative=False, use_efflen_jacobian=True,
                           num_steps=LIKAP_NUM_STEPS, num_mc_samples=LIKAP_NUM_MC_SAMPLES, gradonly=True,
                           seed=123456789, z0=None, gene_transcripts=None):
    """approximate_likelihood(::LogitSkewNormalPTTApprox, sample) (likelihood-approximation.jl:395-624).
    Returns the params Dict: mu, omega, alpha (+ node_parent_idxs, node_js when the tree carries them).
    gene_noninformative = True needs the genes of the transcripts (the reference takes them from the sample's
    transcript metadata, :475-487): `gene_transcripts` = Dict gene id -> 1-based transcript indexes, or an int array
    gene_of[n] (-1 = none known); with no gene information the option is switched off with the reference's warning."""
    if not isinstance(approx, LogitSkewNormalPTTApprox):
        raise NotImplementedError("only LogitSkewNormalPTTApprox is built (the alt approximations are out of scope)")
    if gene_noninformative and gene_transcripts is None:
        gene_transcripts = np.full(sample.n, -1, np.int32)
    if t is None:  # PolyaTreeTransform(X, approx.treemethod) (ptt.jl:35-52, likelihood-approximation.jl:425-440)
        if approx.treemethod == "cluster":
            if getattr(sample, "_csc", None) is None:
                raise ValueError("tree construction needs the sample's CSC arrays (create it from colptr/rowval/nzval) "
                                 "or pass a PolyaTreeTransform")
            parents, js = hclust(sample.m, sample.n, *sample._csc)
        elif approx.treemethod == "cluster_parallel":  # (the rounds variant of the same rule, polee_hclust_parallel)
            if getattr(sample, "_csc", None) is None:
                raise ValueError("tree construction needs the sample's CSC arrays")
            parents, js = hclust(sample.m, sample.n, *sample._csc, parallel=True)
        elif approx.treemethod in ("cluster_device", "cluster_auto"):  # (the same variant built on the GPU, polee_hclust_parallel_device)
            if getattr(sample, "_csc", None) is None:
                raise ValueError("tree construction needs the sample's CSC arrays")
            parents, js = hclust(sample.m, sample.n, *sample._csc, device=True, ctx=sample.ctx)
        elif approx.treemethod == "sequential":
            parents, js = list_nodes(sample.n)
        else:
            raise ValueError("%r is not a supported Polya tree transform heuristic" % (approx.treemethod,))
        t = PolyaTreeTransform(parents, js, ctx=sample.ctx)
    fit = LikelihoodApproximationFit(sample, t, num_steps=num_steps, num_mc_samples=num_mc_samples,
                                     use_efflen_jacobian=use_efflen_jacobian, gradonly=gradonly, seed=seed, z0=z0,
                                     gene_transcripts=gene_transcripts if gene_noninformative else None)
    with _fits_lock:
        _fits_in_flight[0] += 1
    try:
        fit.run(num_steps)
        fit.sync()
    finally:
        with _fits_lock:
            _fits_in_flight[0] -= 1
    mu, omega, alpha = fit.params()
    params = {"mu": mu, "omega": omega, "alpha": alpha}
    if t.node_parent_idxs is not None:
        params["node_parent_idxs"] = t.node_parent_idxs
        params["node_js"] = t.node_js
    if not gradonly:
        params["elbo"], params["lp_mean"] = fit.trace()
    return params


class OptimizePTTApprox:
    """src/likelihood-approximation.jl:4-7."""


def list_nodes(n):
    """Serialised :sequential tree (hclust.jl:477-489 + order_nodes :361-389) -> (node_parent_idxs, node_js).
    It is a caterpillar: leaf n-? ... the k-th merge joins the running subtree (left) with the next leaf (right)."""
    N = 2 * n - 1
    parents = np.zeros(N, np.int32)
    js = np.zeros(N, np.int32)
    # DFS pre-order, right child first: root, its right leaf (transcript 1), then the left subtree, recursively
    idx = 0
    par = 0
    for leaf in range(1, n):  # internal node whose right child is `leaf`
        parents[idx] = par
        me = idx + 1
        idx += 1
        parents[idx] = me
        js[idx] = leaf
        idx += 1
        par = me
    parents[idx] = par
    js[idx] = n
    return parents, js


def optimize_likelihood(sample, t, efflens=None, num_steps=LIKAP_NUM_STEPS):
    """optimize_likelihood / approximate_likelihood(::OptimizePTTApprox, sample)
    (likelihood-approximation.jl:21-23, 149-242) -> {"x": xs}.  `t` is the tree to optimise over (the
    reference uses PolyaTreeTransform(X, :sequential); see list_nodes)."""
    efflens = sample.effective_lengths if efflens is None else efflens
    efflens = arr(efflens, np.float32)
    xs = np.empty(sample.n, np.float32)
    zs = np.empty(sample.n - 1, np.float32)
    check(L.lib().polee_optimize_ptt(sample._h, t._h, ptr(efflens, f32p), int(num_steps), ptr(xs, f32p), ptr(zs, f32p)),
          sample.ctx._h)
    return {"x": xs, "z": zs}


class ApproxLikelihoodSampler:
    """src/approx-sampler.jl:4-44."""

    def __init__(self):
        self.t = self.mu = self.sigma = self.alpha = None
        self._seed = 123456789
        self._count = 0

    def set_transform(self, t, mu, sigma, alpha):
        """set_transform! (approx-sampler.jl:19-34)."""
        self.t = t
        self.mu, self.sigma, self.alpha = (arr(a, np.float32) for a in (mu, sigma, alpha))

    def seed(self, seed):
        self._seed, self._count = int(seed), 0

    def rand(self, ndraws=1, z0=None):
        """rand!(als, xs) (approx-sampler.jl:37-44): ndraws draws -> [ndraws, n] f32."""
        t = self.t
        xs = np.empty((ndraws, t.n), np.float32)
        z = None if z0 is None else arr(z0, np.float32).reshape(ndraws, t.n - 1)
        seed = (self._seed + 0x632BE59BD9B4E019 * self._count) & 0xFFFFFFFFFFFFFFFF
        self._count += 1
        check(L.lib().polee_sampler_draw(t._h, ptr(self.mu, f32p), ptr(self.sigma, f32p), ptr(self.alpha, f32p),
                                         ptr(z, f32p), int(ndraws), C.c_uint64(seed), ptr(xs, f32p)), t.ctx._h)
        return xs

    def _next_seed(self):
        seed = (self._seed + 0x632BE59BD9B4E019 * self._count) & 0xFFFFFFFFFFFFFFFF
        self._count += 1
        return seed

    def posterior_mean(self, N=100, z0=None):
        """One sample of posterior_mean (approx-sampler.jl:86-117): mean of N draws clamped to [1e-15, 0.9999999]."""
        t = self.t
        z = None if z0 is None else arr(z0, np.float32).reshape(N, t.n - 1)
        pm = np.empty(t.n, np.float32)
        check(L.lib().polee_sampler_posterior_mean(t._h, ptr(self.mu, f32p), ptr(self.sigma, f32p),
                                                   ptr(self.alpha, f32p), ptr(z, f32p), int(N),
                                                   C.c_uint64(self._next_seed()), ptr(pm, f32p)), t.ctx._h)
        return pm

    def initial_values(self, efflens, N=30, z0=None):
        """x0 of load_samples_hdf5 (estimate.jl:436-455): mean of N draws with y clamped to [LIKAP_Y_EPS, 1 - LIKAP_Y_EPS],
        each divided by the effective lengths and renormalised."""
        t = self.t
        z = None if z0 is None else arr(z0, np.float32).reshape(N, t.n - 1)
        l = arr(efflens, np.float32).reshape(-1)
        if l.size != t.n:
            raise ValueError("efflens must have one entry per transcript")
        x0 = np.empty(t.n, np.float32)
        check(L.lib().polee_sampler_initial_values(t._h, ptr(self.mu, f32p), ptr(self.sigma, f32p), ptr(self.alpha, f32p),
                                                   ptr(l, f32p), ptr(z, f32p), int(N), C.c_uint64(self._next_seed()),
                                                   ptr(x0, f32p)), t.ctx._h)
        return x0

    def quantile(self, qs=(0.01, 0.99), N=100, z0=None):
        """One sample of Statistics.quantile (approx-sampler.jl:50-83): element-wise quantiles [len(qs), n] of N draws."""
        t = self.t
        z = None if z0 is None else arr(z0, np.float32).reshape(N, t.n - 1)
        q = arr(qs, np.float64).reshape(-1)
        out = np.empty((q.size, t.n), np.float32)
        check(L.lib().polee_sampler_quantiles(t._h, ptr(self.mu, f32p), ptr(self.sigma, f32p), ptr(self.alpha, f32p),
                                              ptr(z, f32p), int(N), C.c_uint64(self._next_seed()), ptr(q, L.f64p),
                                              int(q.size), ptr(out, f32p)), t.ctx._h)
        return out


class RNASeqApproxLikelihood:
    """RNASeqApproxLikelihoodDist (polee_approx_likelihood.py:326-450) for S samples.
    `vars` uses the keys of create_tensorflow_variables! (estimate.jl:502-556):
    efflen [S,n], la_mu/la_sigma/la_alpha [S,n-1], left_index/right_index/leaf_index [S,N]
    (or [1,N] / [N] for a shared tree)."""

    def __init__(self, vars=None, ctx=None, **kw):
        v = dict(vars or {})
        v.update(kw)
        self.ctx = ctx or default_context()
        eff = arr(np.atleast_2d(v["efflen"]), np.float32)
        mu, sg, al = (arr(np.atleast_2d(v[k]), np.float32) for k in ("la_mu", "la_sigma", "la_alpha"))
        li, ri, fi = (arr(np.atleast_2d(v[k]), np.int32) for k in ("left_index", "right_index", "leaf_index"))
        self.S, self.n = eff.shape
        shared = 1 if li.shape[0] == 1 and self.S > 1 else 0
        if not shared and li.shape[0] != self.S:
            raise ValueError("index arrays must have S rows or 1 row")
        self._h = C.c_void_p()
        check(L.lib().polee_approx_create(self.ctx._h, self.S, self.n, ptr(eff, f32p), ptr(mu, f32p), ptr(sg, f32p),
                                          ptr(al, f32p), ptr(li, i32p), ptr(ri, i32p), ptr(fi, i32p), shared,
                                          C.byref(self._h)), self.ctx._h)

    def __del__(self):
        try:
            if self._h:
                L.lib().polee_approx_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def log_prob(self, x, want_grad=False):
        """_log_prob (polee_approx_likelihood.py:367-450): x [S,n] -> lp [S] (and d lp/d x)."""
        x = arr(np.atleast_2d(x), np.float32)
        if x.shape != (self.S, self.n):
            raise ValueError("x must be [%d, %d]" % (self.S, self.n))
        lp = np.empty(self.S, np.float32)
        g = np.empty((self.S, self.n), np.float32) if want_grad else None
        check(L.lib().polee_approx_logprob(self._h, ptr(x, f32p), ptr(lp, f32p), ptr(g, f32p)), self.ctx._h)
        return (lp, g) if want_grad else lp

    def gene_log_prob(self, x_gene, x_isoform, feature_idxs, want_grad=False):
        """RNASeqGeneApproxLikelihoodDist._log_prob (polee_gene_expression.py:14-90): the density reached through
        gene-level expression x_gene [S,G] and within-gene isoform log-expression x_isoform [S,n].
        `feature_idxs` [n]: 1-based gene of every transcript, as the reference passes it (transcript_idxs = 1..n).
        Returns lp [S] (and d lp / d x_gene, d lp / d x_isoform)."""
        xg, xi = arr(np.atleast_2d(x_gene), np.float32), arr(np.atleast_2d(x_isoform), np.float32)
        gene_of = (arr(feature_idxs, np.int64).reshape(-1) - 1).astype(np.int32)
        G = xg.shape[1]
        if xi.shape != (self.S, self.n) or xg.shape[0] != self.S or gene_of.size != self.n:
            raise ValueError("shapes must be x_gene [S,G], x_isoform [S,n], feature_idxs [n]")
        lp = np.empty(self.S, np.float32)
        gg = np.empty((self.S, G), np.float32) if want_grad else None
        gi = np.empty((self.S, self.n), np.float32) if want_grad else None
        check(L.lib().polee_approx_gene_logprob(self._h, ptr(xg, f32p), ptr(xi, f32p), ptr(gene_of, i32p), G,
                                                ptr(lp, f32p), ptr(gg, f32p), ptr(gi, f32p)), self.ctx._h)
        return (lp, gg, gi) if want_grad else lp

    def approximate_feature_likelihood(self, num_features, feature_idxs, transcript_idxs, num_mean_draws=1000,
                                       num_var_draws=1000, seed=123456789, z0=None):
        """approximate_feature_likelihood (polee_gene_expression.py:191-222): normal approximation (loc, scale), each
        [S, num_features], of the log expression of features (sets of transcripts) from sampler draws.
        `feature_idxs`, `transcript_idxs`: equal-length 1-based incidence pairs, as in the reference."""
        fi, ti = arr(feature_idxs, np.int32).reshape(-1), arr(transcript_idxs, np.int32).reshape(-1)
        if fi.size != ti.size:
            raise ValueError("feature_idxs and transcript_idxs must have the same length")
        F = int(num_features)
        z = None
        if z0 is not None:
            z = arr(z0, np.float32).reshape(-1)
            if z.size != (num_mean_draws + num_var_draws) * self.S * (self.n - 1):
                raise ValueError("z0 must hold (num_mean_draws + num_var_draws) x S x (n-1) values")
        loc, scale = np.empty((self.S, F), np.float32), np.empty((self.S, F), np.float32)
        check(L.lib().polee_approx_feature_moments(self._h, ptr(fi, i32p), ptr(ti, i32p), C.c_int64(fi.size), F,
                                                   int(num_mean_draws), int(num_var_draws), C.c_uint64(seed),
                                                   ptr(z, f32p), ptr(loc, f32p), ptr(scale, f32p)), self.ctx._h)
        return loc, scale

    def approximate_splicing_likelihood(self, num_features, feature_indices, antifeature_indices, num_mean_draws=1000,
                                        num_var_draws=1000, seed=123456789, z0=None):
        """approximate_splicing_likelihood (polee_splicing.py:62-113): normal approximation (loc, scale) of the splicing
        log-ratios log(feature) - log(antifeature); index arrays [P,2] / [Q,2] of (feature, transcript), 0-based."""
        fi, afi = arr(feature_indices, np.int32).reshape(-1, 2), arr(antifeature_indices, np.int32).reshape(-1, 2)
        F = int(num_features)
        z = None if z0 is None else arr(z0, np.float32).reshape(-1)
        if z is not None and z.size != (num_mean_draws + num_var_draws) * self.S * (self.n - 1):
            raise ValueError("z0 must hold (num_mean_draws + num_var_draws) x S x (n-1) values")
        loc, scale = np.empty((self.S, F), np.float32), np.empty((self.S, F), np.float32)
        check(L.lib().polee_approx_splicing_moments(self._h, ptr(fi, i32p), C.c_int64(fi.shape[0]), ptr(afi, i32p),
                                                    C.c_int64(afi.shape[0]), F, int(num_mean_draws), int(num_var_draws),
                                                    C.c_uint64(seed), ptr(z, f32p), ptr(loc, f32p), ptr(scale, f32p)),
              self.ctx._h)
        return loc, scale

    def sample(self, z0=None, seed=123456789):
        """rnaseq_approx_likelihood_sampler (polee_approx_likelihood.py:35-59), one draw per sample."""
        z = None if z0 is None else arr(z0, np.float32).reshape(self.S, self.n - 1)
        x = np.empty((self.S, self.n), np.float32)
        check(L.lib().polee_approx_sample(self._h, ptr(z, f32p), C.c_uint64(seed), ptr(x, f32p)), self.ctx._h)
        return x


def rnaseq_approx_likelihood_sampler(vars, z0=None, seed=123456789, ctx=None):
    return RNASeqApproxLikelihood(vars, ctx=ctx).sample(z0, seed)
