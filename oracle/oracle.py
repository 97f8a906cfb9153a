"""ctypes wrapper of oracle/polee_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of polee_oracle.c).  Parity pin status: "partially
pinned" (reference fixtures only; no reference-produced numeric vectors exist).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpolee_oracle.so")
_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "polee_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_ptt_create.restype = C.c_void_p
        _lib.oracle_sample_create.restype = C.c_void_p
        _lib.oracle_ptt_index.restype = c_i32p
        _lib.oracle_ptt_us.restype = c_f64p
        _lib.oracle_sample_frag_probs.restype = c_f64p
        _lib.oracle_sample_tcolptr.restype = c_u64p
        _lib.oracle_sample_trowval.restype = c_u32p
        _lib.oracle_sample_tnzval.restype = c_f32p
        for f in ("oracle_ptt_transform", "oracle_ptt_inverse_transform", "oracle_log_likelihood",
                  "oracle_factored_log_likelihood", "oracle_effective_length_jacobian_adjustment",
                  "oracle_kumaraswamy_transform", "oracle_adam_learning_rate"):
            getattr(_lib, f).restype = C.c_double
        for f in ("oracle_logit_normal_transform", "oracle_sinh_asinh_transform"):
            getattr(_lib, f).restype = C.c_float
    return _lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class PTT:
    """PolyaTreeTransform (src/ptt.jl:6-27, 89-116)."""

    def __init__(self, node_parent_idxs, node_js):
        self.parent = _i32(node_parent_idxs)
        self.js = _i32(node_js)
        self.N = int(self.parent.size)
        self.n = (self.N + 1) // 2
        self.h = C.c_void_p(lib().oracle_ptt_create(_p(self.parent, c_i32p), _p(self.js, c_i32p), self.N))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.oracle_ptt_destroy(self.h)
            self.h = None

    @property
    def index(self):
        """4 x N int32 (rows: leaf id, left, right, parent; 1-based, 0 = none)."""
        ptr = lib().oracle_ptt_index(self.h)
        return np.ctypeslib.as_array(ptr, shape=(self.N, 4)).T.copy()

    @property
    def us(self):
        return np.ctypeslib.as_array(lib().oracle_ptt_us(self.h), shape=(self.N,)).copy()

    def transform(self, ys, compute_ladj=False):
        ys = _f64(ys)
        xs = np.empty(self.n, np.float32)
        ladj = lib().oracle_ptt_transform(self.h, _p(ys, c_f64p), _p(xs, c_f32p), int(compute_ladj))
        return xs, ladj

    def transform_gradients(self, ys, x_grad):
        ys, x_grad = _f64(ys), _f64(x_grad)
        y_grad = np.zeros(self.n - 1, np.float32)
        lib().oracle_ptt_transform_gradients(self.h, _p(ys, c_f64p), _p(y_grad, c_f32p), _p(x_grad, c_f64p))
        return y_grad

    def transform_gradients_no_ladj(self, ys, x_grad):
        ys, x_grad = _f64(ys), _f64(x_grad)
        y_grad = np.zeros(self.n - 1, np.float64)
        lib().oracle_ptt_transform_gradients_no_ladj(self.h, _p(ys, c_f64p), _p(y_grad, c_f64p),
                                                     _p(x_grad, c_f64p))
        return y_grad

    def inverse_transform(self, xs):
        xs = _f32(xs)
        ys = np.empty(self.n - 1, np.float64)
        ladj = lib().oracle_ptt_inverse_transform(self.h, _p(xs, c_f32p), _p(ys, c_f64p))
        return ys, ladj


def make_inverse_ptt_params(node_parent_idxs, node_js):
    p, j = _i32(node_parent_idxs), _i32(node_js)
    N = p.size
    l, r, f = (np.empty(N, np.int32) for _ in range(3))
    lib().oracle_make_inverse_ptt_params(_p(p, c_i32p), _p(j, c_i32p), N, _p(l, c_i32p), _p(r, c_i32p),
                                         _p(f, c_i32p))
    return l, r, f


def _tree_args(left, right, leaf):
    left, right, leaf = _i32(left), _i32(right), _i32(leaf)
    shared = 1 if left.ndim == 1 or left.shape[0] == 1 else 0
    return left, right, leaf, shared


def hsb(y_logit, left, right, leaf):
    y_logit = _f32(np.atleast_2d(y_logit))
    B, nm1 = y_logit.shape
    left, right, leaf, shared = _tree_args(left, right, leaf)
    x = np.empty((B, nm1 + 1), np.float32)
    lib().oracle_hsb(_p(y_logit, c_f32p), _p(left, c_i32p), _p(right, c_i32p), _p(leaf, c_i32p),
                     C.c_int64(B), C.c_int64(nm1 + 1), shared, _p(x, c_f32p))
    return x


def inv_hsb(x, left, right, leaf):
    x = _f32(np.atleast_2d(x))
    B, n = x.shape
    left, right, leaf, shared = _tree_args(left, right, leaf)
    y = np.empty((B, n - 1), np.float64)
    ladj = np.empty((B, 1), np.float32)
    lib().oracle_inv_hsb(_p(x, c_f32p), _p(left, c_i32p), _p(right, c_i32p), _p(leaf, c_i32p),
                         C.c_int64(B), C.c_int64(n), shared, _p(y, c_f64p), _p(ladj, c_f32p))
    return y, ladj


def inv_hsb_grad(y_grad, ladj_grad, y, left, right, leaf):
    y_grad, y = _f64(np.atleast_2d(y_grad)), _f64(np.atleast_2d(y))
    ladj_grad = _f32(np.reshape(ladj_grad, (-1,)))
    B, nm1 = y.shape
    left, right, leaf, shared = _tree_args(left, right, leaf)
    bp = np.empty((B, nm1 + 1), np.float32)
    lib().oracle_inv_hsb_grad(_p(y_grad, c_f64p), _p(ladj_grad, c_f32p), _p(y, c_f64p), _p(left, c_i32p),
                              _p(right, c_i32p), _p(leaf, c_i32p), C.c_int64(B), C.c_int64(nm1 + 1), shared,
                              _p(bp, c_f32p))
    return bp


class Sample:
    """X (CSC m x n, 1-based colptr/rowval as in the likelihood-matrix HDF5) + Model scratch."""

    def __init__(self, m, n, colptr, rowval, nzval):
        self.m, self.n = int(m), int(n)
        colptr = np.ascontiguousarray(colptr)
        if colptr.dtype not in (np.uint32, np.uint64):
            colptr = colptr.astype(np.uint64)
        rowval = np.ascontiguousarray(rowval, dtype=np.uint32)
        nzval = _f32(nzval)
        self.nnz = int(colptr[-1]) - 1
        self.h = C.c_void_p(lib().oracle_sample_create(
            C.c_int64(self.m), C.c_int64(self.n), colptr.ctypes.data_as(C.c_void_p), colptr.dtype.itemsize,
            _p(rowval, c_u32p), _p(nzval, c_f32p)))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.oracle_sample_destroy(self.h)
            self.h = None

    def csr(self):
        """(rowptr u64[m+1] 1-based, col u32[nnz] 1-based, val f32[nnz]) = Xt in CSC form."""
        L = lib()
        tp = np.ctypeslib.as_array(L.oracle_sample_tcolptr(self.h), shape=(self.m + 1,)).copy()
        tr = np.ctypeslib.as_array(L.oracle_sample_trowval(self.h), shape=(self.nnz,)).copy()
        tv = np.ctypeslib.as_array(L.oracle_sample_tnzval(self.h), shape=(self.nnz,)).copy()
        return tp, tr, tv

    @property
    def frag_probs(self):
        return np.ctypeslib.as_array(lib().oracle_sample_frag_probs(self.h), shape=(self.m,)).copy()

    def log_likelihood(self, xs, gradonly=False):
        xs = _f32(xs)
        x_grad = np.zeros(self.n, np.float64)
        lp = lib().oracle_log_likelihood(self.h, _p(xs, c_f32p), _p(x_grad, c_f64p), int(gradonly))
        return lp, x_grad

    def factored_log_likelihood(self, ks, xs, gradonly=False):
        xs = _f32(xs)
        ks = np.ascontiguousarray(ks, dtype=np.int64)
        x_grad = np.zeros(self.n, np.float64)
        lp = lib().oracle_factored_log_likelihood(self.h, _p(ks, c_i64p), _p(xs, c_f32p), _p(x_grad, c_f64p),
                                                  int(gradonly))
        return lp, x_grad


def effective_length_jacobian_adjustment(efflens, xs, x_grad):
    efflens, xs = _f32(efflens), _f32(xs)
    x_grad = _f64(x_grad).copy()
    xls = np.empty_like(xs)
    lib().oracle_effective_length_jacobian_adjustment(_p(efflens, c_f32p), _p(xs, c_f32p), _p(xls, c_f32p),
                                                      _p(x_grad, c_f64p), C.c_int64(xs.size))
    return xls, x_grad


def gene_noninformative_prior(efflens, xls, xs, x_grad, gene_of):
    """likelihood.jl:114-159; gene_of int32[n], -1 = no gene.  Returns the adjusted x_grad."""
    efflens, xls, xs = _f32(efflens), _f32(xls), _f32(xs)
    gene_of = np.ascontiguousarray(gene_of, np.int32)
    x_grad = _f64(x_grad).copy()
    f = lib().oracle_gene_noninformative_prior
    f.restype = C.c_double
    f(_p(efflens, c_f32p), _p(xls, c_f32p), _p(xs, c_f32p), _p(x_grad, c_f64p),
      gene_of.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(xs.size), C.c_int64(int(gene_of.max(initial=-1)) + 1))
    return x_grad


def sinh_asinh_transform(alpha, zs0, compute_ladj=False):
    alpha, zs0 = _f32(alpha), _f32(zs0)
    zs = np.empty_like(zs0)
    ladj = lib().oracle_sinh_asinh_transform(_p(alpha, c_f32p), _p(zs0, c_f32p), _p(zs, c_f32p),
                                             C.c_int64(zs0.size), int(compute_ladj))
    return zs, ladj


def logit_normal_transform(mu, sigma, zs, compute_ladj=False):
    mu, sigma, zs = _f32(mu), _f32(sigma), _f32(zs)
    ys = np.empty(zs.size, np.float64)
    ladj = lib().oracle_logit_normal_transform(_p(mu, c_f32p), _p(sigma, c_f32p), _p(zs, c_f32p), _p(ys, c_f64p),
                                               C.c_int64(zs.size), int(compute_ladj))
    return ys, ladj


def logit_normal_transform_gradients(zs, ys, mu, sigma, y_grad):
    zs, mu, sigma, y_grad = _f32(zs), _f32(mu), _f32(sigma), _f32(y_grad)
    ys = _f64(ys)
    z_grad, mu_grad, sigma_grad = (np.zeros(zs.size, np.float32) for _ in range(3))
    lib().oracle_logit_normal_transform_gradients(
        _p(zs, c_f32p), _p(ys, c_f64p), _p(mu, c_f32p), _p(sigma, c_f32p), _p(y_grad, c_f32p),
        _p(z_grad, c_f32p), _p(mu_grad, c_f32p), _p(sigma_grad, c_f32p), C.c_int64(zs.size))
    return z_grad, mu_grad, sigma_grad


def sinh_asinh_transform_gradients(zs0, alpha, z_grad):
    zs0, alpha, z_grad = _f32(zs0), _f32(alpha), _f32(z_grad)
    alpha_grad = np.zeros(zs0.size, np.float32)
    lib().oracle_sinh_asinh_transform_gradients(_p(zs0, c_f32p), _p(alpha, c_f32p), _p(z_grad, c_f32p),
                                                _p(alpha_grad, c_f32p), C.c_int64(zs0.size))
    return alpha_grad


def kumaraswamy_transform(a, b, zs, compute_ladj=True):
    a, b, zs = _f32(a), _f32(b), _f32(zs)
    ys = np.empty(zs.size, np.float64)
    ladj = lib().oracle_kumaraswamy_transform(_p(a, c_f32p), _p(b, c_f32p), _p(zs, c_f32p), _p(ys, c_f64p),
                                              C.c_int64(zs.size), int(compute_ladj))
    return ys, ladj


def kumaraswamy_transform_gradients(zs, a, b, y_grad):
    zs, a, b, y_grad = _f32(zs), _f32(a), _f32(b), _f32(y_grad)
    a_grad, b_grad = np.zeros(zs.size, np.float32), np.zeros(zs.size, np.float32)
    lib().oracle_kumaraswamy_transform_gradients(_p(zs, c_f32p), _p(a, c_f32p), _p(b, c_f32p), _p(y_grad, c_f32p),
                                                 _p(a_grad, c_f32p), _p(b_grad, c_f32p), C.c_int64(zs.size))
    return a_grad, b_grad


def adam_learning_rate(step_num):
    return lib().oracle_adam_learning_rate(C.c_double(step_num))


def randn(n, seed):
    out = np.empty(int(n), np.float32)
    lib().oracle_randn_fill(_p(out, c_f32p), C.c_int64(out.size), C.c_uint64(seed))
    return out


def approximate_likelihood(sample, ptt, efflens, num_steps=500, num_mc=6, use_efflen_jacobian=True,
                           gradonly=True, z0=None, seed=123456789, ks=None, init_only=False, gene_of=None):
    """src/likelihood-approximation.jl:395-575. Returns dict(mu, omega, alpha[, elbo, lp_mean]).
    gene_of (int32[n], -1 = no gene known): gene_noninformative = true (:475-491, 535-538)."""
    efflens = _f32(efflens)
    nm1 = sample.n - 1
    mu, omega, alpha = (np.zeros(nm1, np.float32) for _ in range(3))
    elbo = np.zeros(num_steps, np.float64)
    lpm = np.zeros(num_steps, np.float64)
    if z0 is not None:
        z0 = _f32(z0)
        assert z0.size == num_steps * num_mc * nm1
    if ks is not None:
        ks = np.ascontiguousarray(ks, dtype=np.int64)
    ng = 0
    if gene_of is not None:
        gene_of = np.ascontiguousarray(gene_of, np.int32)
        assert gene_of.size == sample.n
        ng = int(gene_of.max(initial=-1)) + 1
    rc = lib().oracle_approximate_likelihood(
        sample.h, ptt.h, _p(efflens, c_f32p), _p(ks, c_i64p), int(num_steps), int(num_mc),
        int(use_efflen_jacobian), int(gradonly), _p(z0, c_f32p), C.c_uint64(seed), int(init_only),
        _p(mu, c_f32p), _p(omega, c_f32p), _p(alpha, c_f32p), _p(elbo, c_f64p), _p(lpm, c_f64p),
        _p(gene_of, c_i32p), C.c_int64(ng))
    if rc != 0:
        raise FloatingPointError("non-finite gradient (likelihood-approximation.jl:559)")
    return dict(mu=mu, omega=omega, alpha=alpha, elbo=elbo, lp_mean=lpm)


def optimize_ptt(sample, ptt, efflens, num_steps=500):
    """approximate_likelihood(::OptimizePTTApprox, sample) (likelihood-approximation.jl:149-242) -> (xs, zs)."""
    efflens = _f32(efflens)
    xs = np.empty(sample.n, np.float32)
    zs = np.empty(sample.n - 1, np.float32)
    lib().oracle_optimize_ptt(sample.h, ptt.h, _p(efflens, c_f32p), int(num_steps), _p(zs, c_f32p), _p(xs, c_f32p))
    return xs, zs


def list_nodes(n):
    """The :sequential tree (hclust.jl:477-489 list_nodes + order_nodes :361-389), serialised."""
    stack = list(range(1, n + 1))
    while len(stack) > 1:
        a = stack.pop(); b = stack.pop()
        stack.append((a, b))  # HClustNode(a, b): left = a, right = b
    parents, js = [], []
    st = [(stack[0], 0)]
    while st:
        node, par = st.pop()
        idx = len(parents) + 1
        parents.append(par)
        if isinstance(node, tuple):
            js.append(0)
            st.append((node[0], idx)); st.append((node[1], idx))  # right is popped (visited) first
        else:
            js.append(int(node))
    return np.array(parents, np.int32), np.array(js, np.int32)


def vi_draw_gradients(sample, ptt, efflens, mu, omega, alpha, zs0, use_efflen_jacobian=True):
    efflens, mu, omega, alpha, zs0 = map(_f32, (efflens, mu, omega, alpha, zs0))
    n, nm1 = sample.n, sample.n - 1
    xs = np.empty(n, np.float32)
    x_grad = np.empty(n, np.float64)
    y_grad, mu_g, om_g, al_g = (np.empty(nm1, np.float32) for _ in range(4))
    lp, ladj = C.c_double(), C.c_double()
    ys = np.empty(nm1, np.float64)
    lib().oracle_vi_draw_gradients(
        sample.h, ptt.h, _p(efflens, c_f32p), int(use_efflen_jacobian), _p(mu, c_f32p), _p(omega, c_f32p),
        _p(alpha, c_f32p), _p(zs0, c_f32p), _p(xs, c_f32p), _p(x_grad, c_f64p), _p(y_grad, c_f32p),
        _p(mu_g, c_f32p), _p(om_g, c_f32p), _p(al_g, c_f32p), C.byref(lp), C.byref(ladj), _p(ys, c_f64p))
    return dict(ys=ys, xs=xs, x_grad=x_grad, y_grad=y_grad, mu_grad=mu_g, omega_grad=om_g, alpha_grad=al_g,
                lp=lp.value, ladj=ladj.value)


def set_debug_scale(term, scale):
    """Scale one of the six additive gradient terms of the ELBO (see polee_oracle.c: oracle_debug_scale); 1.0 = off."""
    lib().oracle_set_debug_scale(int(term), C.c_double(scale))


def vi_pin_stats(sample, ptt, efflens, mu, omega, alpha, seed0, ndraws):
    """(mean, standard error) [3, n-1] of the per-draw gradients of vi_draw_gradients over draws seed0 .. seed0+ndraws-1."""
    efflens, mu, omega, alpha = map(_f32, (efflens, mu, omega, alpha))
    nm1 = sample.n - 1
    gsum, gsq = np.empty((3, nm1), np.float64), np.empty((3, nm1), np.float64)
    lib().oracle_vi_pin_stats(sample.h, ptt.h, _p(efflens, c_f32p), _p(mu, c_f32p), _p(omega, c_f32p), _p(alpha, c_f32p),
                              C.c_uint64(seed0), C.c_int64(ndraws), _p(gsum, c_f64p), _p(gsq, c_f64p))
    mean = gsum / ndraws
    var = np.maximum(gsq / ndraws - mean ** 2, 0.0)
    return mean, np.sqrt(var / max(ndraws - 1, 1))


def sampler_draw(ptt, mu, sigma, alpha, zs0):
    mu, sigma, alpha, zs0 = map(_f32, (mu, sigma, alpha, zs0))
    xs = np.empty(ptt.n, np.float32)
    lib().oracle_sampler_draw(ptt.h, _p(mu, c_f32p), _p(sigma, c_f32p), _p(alpha, c_f32p), _p(zs0, c_f32p),
                              _p(xs, c_f32p))
    return xs


def x0_draw(ptt, mu, sigma, alpha, efflens, zs0):
    mu, sigma, alpha, efflens, zs0 = map(_f32, (mu, sigma, alpha, efflens, zs0))
    x0 = np.empty(ptt.n, np.float32)
    lib().oracle_x0_draw(ptt.h, _p(mu, c_f32p), _p(sigma, c_f32p), _p(alpha, c_f32p), _p(efflens, c_f32p),
                         _p(zs0, c_f32p), _p(x0, c_f32p))
    return x0


def tf_sampler(z0, efflens, mu, sigma, alpha, left, right, leaf):
    z0, efflens, mu, sigma, alpha = (_f32(np.atleast_2d(a)) for a in (z0, efflens, mu, sigma, alpha))
    S, nm1 = z0.shape
    left, right, leaf, shared = _tree_args(left, right, leaf)
    x = np.empty((S, nm1 + 1), np.float32)
    lib().oracle_tf_sampler(_p(z0, c_f32p), _p(efflens, c_f32p), _p(mu, c_f32p), _p(sigma, c_f32p),
                            _p(alpha, c_f32p), _p(left, c_i32p), _p(right, c_i32p), _p(leaf, c_i32p),
                            C.c_int64(S), C.c_int64(nm1 + 1), shared, _p(x, c_f32p))
    return x


def approx_log_prob(x, efflens, mu, sigma, alpha, left, right, leaf, want_grad=False):
    """polee_approx_likelihood.py:367-450 for one leading index; x [S, n]."""
    x, efflens, mu, sigma, alpha = (_f32(np.atleast_2d(a)) for a in (x, efflens, mu, sigma, alpha))
    S, n = x.shape
    left, right, leaf, shared = _tree_args(left, right, leaf)
    lp = np.empty(S, np.float32)
    g = np.empty((S, n), np.float32) if want_grad else None
    lib().oracle_approx_log_prob(_p(x, c_f32p), _p(efflens, c_f32p), _p(mu, c_f32p), _p(sigma, c_f32p),
                                 _p(alpha, c_f32p), _p(left, c_i32p), _p(right, c_i32p), _p(leaf, c_i32p),
                                 C.c_int64(S), C.c_int64(n), shared, _p(lp, c_f32p), _p(g, c_f32p))
    return (lp, g) if want_grad else lp


def approx_gene_log_prob(x_gene, x_iso, gene_of, efflens, mu, sigma, alpha, left, right, leaf, want_grad=False):
    """polee_gene_expression.py:14-90 around approx_log_prob; x_gene [S, G], x_iso [S, n]."""
    x_gene, x_iso, efflens, mu, sigma, alpha = (_f32(np.atleast_2d(a)) for a in (x_gene, x_iso, efflens, mu, sigma, alpha))
    S, n = x_iso.shape
    G = x_gene.shape[1]
    gene_of = np.ascontiguousarray(gene_of, np.int32)
    left, right, leaf, shared = _tree_args(left, right, leaf)
    lp = np.empty(S, np.float32)
    gg = np.empty((S, G), np.float32) if want_grad else None
    gi = np.empty((S, n), np.float32) if want_grad else None
    f = lib().oracle_approx_gene_log_prob
    f.restype = None
    f(_p(x_gene, c_f32p), _p(x_iso, c_f32p), _p(gene_of, c_i32p), C.c_int64(G), _p(efflens, c_f32p), _p(mu, c_f32p),
      _p(sigma, c_f32p), _p(alpha, c_f32p), _p(left, c_i32p), _p(right, c_i32p), _p(leaf, c_i32p), C.c_int64(S),
      C.c_int64(n), shared, _p(lp, c_f32p), _p(gg, c_f32p), _p(gi, c_f32p))
    return (lp, gg, gi) if want_grad else lp


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(t):
    lib().oracle_set_num_threads(int(t))


def physical_cores():
    """Physical cores of this host (distinct (socket, core) pairs of /proc/cpuinfo): the reference's default thread
    count (polee:8-12)."""
    pairs, phys, n = set(), "0", 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                n += 1
            elif line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
    except OSError:
        pass
    cores = len(pairs) or n or (os.cpu_count() or 1)
    # a container may be given fewer CPUs than the machine has (cgroup CPU quota): threads beyond it are only throttled
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    q = cpu_quota()
    return max(1, min(cores, q)) if q else cores


def cpu_quota():
    """CPUs this process may use according to its cgroup's CPU quota (cgroup v2 cpu.max / v1 cfs quota), rounded up;
    None when unlimited."""
    try:
        a = open("/sys/fs/cgroup/cpu.max").read().split()
        if a and a[0] != "max":
            return max(1, -(-int(a[0]) // int(a[1])))
        return None
    except (OSError, ValueError, IndexError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, -(-quota // period)) if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None
