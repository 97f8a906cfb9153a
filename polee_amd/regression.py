"""Host-side mirror of the regression model (models/polee_regression.py) over libpolee_hip's polee_regression_*.

Same class names, constructor arguments and return values as the reference; the TensorFlow-Probability
joint distributions and fit_surrogate_posterior are replaced by the device step in csrc/regression.hip.
Also: estimate_sample_scales (src/PoleeModel.jl:82-89) and the effect-size output semantics
(src/regression.jl:604-685).
"""
import ctypes as C
import math

import numpy as np

from . import _lib as L
from ._lib import arr, check, f32p, ptr
from .core import RNASeqApproxLikelihood, default_context

# (name, shape code) of the flat parameter vector, in include/polee_hip.h's order
PARAM_TABLE = [
    ("qw_global_scale_variance_loc", "1"), ("qw_global_scale_variance_softplus_scale", "1"),
    ("qw_global_scale_noncentered_loc", "1"), ("qw_global_scale_noncentered_softplus_scale", "1"),
    ("qw_distortion_c_loc", "Fd"), ("qx_scale_concentration_c_loc", "d"), ("qx_scale_scale_c_loc", "d"),
    ("qw_local1_scale_variance_loc", "Fn"), ("qw_local1_scale_variance_softplus_scale", "Fn"),
    ("qw_local1_scale_noncentered_loc", "Fn"), ("qw_local1_scale_noncentered_softplus_scale", "Fn"),
    ("qw_local2_scale_variance_loc", "Fn"), ("qw_local2_scale_variance_softplus_scale", "Fn"),
    ("qw_local2_scale_noncentered_loc", "Fn"), ("qw_local2_scale_noncentered_softplus_scale", "Fn"),
    ("qw_loc", "Fn"), ("qw_softplus_scale", "Fn"),
    ("qx_bias_loc", "n"), ("qx_bias_softplus_scale", "n"),
    ("qx_scale_loc", "n"), ("qx_scale_softplus_scale", "n"),
    ("qx_loc", "Sn"), ("qx_softplus_scale", "Sn"),
]


def _softmax(a):
    e = np.exp(a - a.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def relaxed_onehot_terms(logits, y, T):
    """RelaxedOneHotCategorical(temperature T, logits) at y on the simplex (Maddison et al. 2017, eq. 10; TFP builds it as
    Exp(ExpRelaxedOneHotCategorical)):  log q(y) = lgamma(K) + (K - 1) log T + sum_k (logits_k - (T + 1) log y_k)
    - K logsumexp_k(logits_k - T log y_k).  Returns (log q [S], d log q / d logits at fixed y [S, K], d log q / d y [S, K])."""
    logits, y = np.asarray(logits, np.float64), np.asarray(y, np.float64)
    K = logits.shape[1]
    ly = np.log(y)
    u = logits - T * ly
    mx = u.max(axis=1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(u - mx).sum(axis=1))
    sm = np.exp(u - lse[:, None])
    lq = math.lgamma(K) + (K - 1) * math.log(T) + (logits - (T + 1.0) * ly).sum(axis=1) - K * lse
    return lq, 1.0 - K * sm, (-(T + 1.0) + K * T * sm) / y


def _softplus(x):
    return np.logaddexp(0.0, x).astype(np.float32)


def estimate_sample_scales(x, upper_quantile=0.95):
    """src/PoleeModel.jl:82-89: per-sample offsets from the highly expressed features; x [S, n] log expression.
    Returns [S, 1] like the reference."""
    x = np.asarray(x, np.float64)
    x_mean = np.median(x, axis=0)
    high = x_mean > np.quantile(x_mean, upper_quantile)
    return np.median(x_mean[high][None, :] - x[:, high], axis=1, keepdims=True).astype(np.float32)


class RNASeqLinearRegression:
    """RNASeqLinearRegression (models/polee_regression.py:18-340).  `likelihood_model` is an
    RNASeqApproxLikelihood (or None with point estimates) instead of a TFP coroutine."""

    def __init__(self, F, x_init, likelihood_model, x_bias_loc0, x_bias_scale0, x_scale_hinges, sample_scales,
                 use_distortion, scale_penalty, use_point_estimates, kernel_regression_degree,
                 kernel_regression_bandwidth, ctx=None, comm=None, x_init_mean=None, normal_likelihood=None,
                 gene_likelihood=None):
        """comm / x_init_mean: samples sharded over ranks (polee_regression_set_comm) -- F, x_init, sample_scales are
        this rank's rows, x_init_mean the column means of x_init over all samples."""
        Fm = arr(np.atleast_2d(F), np.float32)
        x0 = arr(np.atleast_2d(x_init), np.float32)
        ss = arr(np.asarray(sample_scales).reshape(-1), np.float32)
        self.num_samples, self.num_factors = Fm.shape
        self.design = Fm
        self.num_features = x0.shape[1]
        if x0.shape[0] != self.num_samples or ss.size != self.num_samples:
            raise ValueError("F [S,F], x_init [S,n] and sample_scales [S] disagree on S")
        self.kernel_regression_degree = int(kernel_regression_degree)
        hg = None if x_scale_hinges is None else arr(np.asarray(x_scale_hinges).reshape(-1), np.float32)
        if hg is not None and hg.size != self.kernel_regression_degree:
            raise ValueError("x_scale_hinges must hold kernel_regression_degree values")
        self.use_point_estimates = bool(use_point_estimates)
        self.likelihood_model = likelihood_model
        # (kept for classify(): the testing model is created with the training model's settings)
        self._ctor = dict(x_bias_loc0=float(x_bias_loc0), x_bias_scale0=float(x_bias_scale0), use_distortion=bool(use_distortion),
                          scale_penalty=float(scale_penalty), kernel_regression_bandwidth=float(kernel_regression_bandwidth),
                          x_scale_hinges=None if hg is None else hg.copy(),
                          x_init_mean=(x0.astype(np.float64).mean(axis=0) if x_init_mean is None
                                       else np.asarray(x_init_mean, np.float64).reshape(-1)))
        if (not self.use_point_estimates and likelihood_model is None and normal_likelihood is None
                and gene_likelihood is None):
            raise ValueError("a likelihood model is needed unless use_point_estimates")
        if gene_likelihood is not None and ctx is None:
            ctx = gene_likelihood[0].ctx
        self.ctx = ctx or (likelihood_model.ctx if likelihood_model is not None else default_context())
        self._h = C.c_void_p()
        ap = likelihood_model._h if (likelihood_model is not None and not self.use_point_estimates) else None
        xm = None if x_init_mean is None else arr(np.asarray(x_init_mean).reshape(-1), np.float32)
        if xm is not None and xm.size != self.num_features:
            raise ValueError("x_init_mean must hold one value per feature")
        check(L.lib().polee_regression_create(
            self.ctx._h, ap, self.num_samples, self.num_factors, self.num_features, ptr(Fm, f32p), ptr(x0, f32p),
            ptr(xm, f32p), ptr(ss, f32p), ptr(hg, f32p), self.kernel_regression_degree, C.c_float(kernel_regression_bandwidth),
            C.c_float(x_bias_loc0), C.c_float(x_bias_scale0), int(bool(use_distortion)), C.c_float(scale_penalty),
            int(self.use_point_estimates), C.byref(self._h)), self.ctx._h)
        lib = L.lib()
        lib.polee_regression_num_params.restype = C.c_int64
        lib.polee_regression_num_noise.restype = C.c_int64
        lib.polee_regression_num_params.argtypes = [C.c_void_p]
        lib.polee_regression_num_noise.argtypes = [C.c_void_p]
        self.num_params = int(lib.polee_regression_num_params(self._h))
        self.num_noise = int(lib.polee_regression_num_noise(self._h))
        if normal_likelihood is not None:
            loc, scale = (arr(np.atleast_2d(a), np.float32) for a in normal_likelihood)
            if loc.shape != x0.shape or scale.shape != x0.shape:
                raise ValueError("the Normal likelihood's loc and scale must be [S, n]")
            check(lib.polee_regression_set_normal_likelihood(self._h, ptr(loc, f32p), ptr(scale, f32p)), self.ctx._h)
        self.num_isoform_params = 0
        if gene_likelihood is not None:  # (RNASeqApproxLikelihood over the transcripts, 0-based gene of each, init
            # [, isoform design: the gene-isoform model])
            lik, gene_of, xi0 = gene_likelihood[:3]
            Fi = arr(np.atleast_2d(gene_likelihood[3]), np.float32) if len(gene_likelihood) > 3 else None
            gene_of, xi0 = arr(gene_of, np.int32).reshape(-1), arr(np.atleast_2d(xi0), np.float32)
            if gene_of.size != lik.n or xi0.shape != (self.num_samples, lik.n):
                raise ValueError("gene_of must be [nt] and x_isoform_init [S, nt]")
            self.likelihood_model = lik
            if Fi is not None:
                if Fi.shape[0] != self.num_samples:
                    raise ValueError("F_isoform must be [S, Fi]")
                self.num_isoform_factors = Fi.shape[1]
                check(lib.polee_regression_set_gene_isoform_likelihood(
                    self._h, lik._h, ptr(gene_of, L.i32p), ptr(xi0, f32p), ptr(Fi, f32p), Fi.shape[1]), self.ctx._h)
            else:
                check(lib.polee_regression_set_gene_likelihood(self._h, lik._h, ptr(gene_of, L.i32p), ptr(xi0, f32p)),
                      self.ctx._h)
            lib.polee_regression_num_isoform_params.restype = C.c_int64
            lib.polee_regression_num_isoform_params.argtypes = [C.c_void_p]
            self.num_isoform_params = int(lib.polee_regression_num_isoform_params(self._h))
            self.num_noise = int(lib.polee_regression_num_noise(self._h))
        self.comm = comm
        if comm is not None:
            check(lib.polee_regression_set_comm(self._h, comm._h), self.ctx._h)

    def __del__(self):
        try:
            if self._h:
                f = L.lib().polee_regression_destroy
                f.restype = None
                f.argtypes = [C.c_void_p]
                f(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- parameters by the reference's variable names
    def _shape(self, code):
        S, F, n, d = self.num_samples, self.num_factors, self.num_features, self.kernel_regression_degree
        return {"1": (), "Fd": (F, d), "d": (d,), "Fn": (F, n), "n": (n,), "Sn": (S, n)}[code]

    def get_flat_params(self):
        p = np.empty(self.num_params, np.float32)
        check(L.lib().polee_regression_get_params(self._h, ptr(p, f32p)), self.ctx._h)
        return p

    def set_flat_params(self, p):
        p = arr(p, np.float32).reshape(-1)
        if p.size != self.num_params:
            raise ValueError("expected %d parameters" % self.num_params)
        check(L.lib().polee_regression_set_params(self._h, ptr(p, f32p)), self.ctx._h)

    def unflatten(self, vec):
        out, o = {}, 0
        for name, code in PARAM_TABLE:
            shp = self._shape(code)
            k = int(np.prod(shp)) if shp else 1
            out[name] = vec[o:o + k].reshape(shp)
            o += k
        return out

    def variables(self):
        """All surrogate-posterior variables (the reference's `<name>_var`) as a dict of arrays."""
        return self.unflatten(self.get_flat_params())

    def kernel_regression_weights(self):
        w = np.empty((self.kernel_regression_degree, self.num_features), np.float32)
        check(L.lib().polee_regression_weights(self._h, ptr(w, f32p)), self.ctx._h)
        return w

    def get_x_posterior_params(self):
        """models/polee_regression.py:121-122"""
        v = self.variables()
        return v["qx_loc"], _softplus(v["qx_softplus_scale"])

    def loss_and_gradients(self, noise=None, seed=123456789):
        """One evaluation of the variational loss and its gradient (flat), no update."""
        z = None if noise is None else arr(noise, np.float32).reshape(-1)
        if z is not None and z.size != self.num_noise:
            raise ValueError("expected %d noise values" % self.num_noise)
        loss = np.empty(1, np.float32)
        g = np.empty(self.num_params, np.float32)
        check(L.lib().polee_regression_eval(self._h, ptr(z, f32p), C.c_uint64(seed), ptr(loss, f32p), ptr(g, f32p)),
              self.ctx._h)
        return float(loss[0]), g

    # ---- isoform block of the gene-level model
    def get_isoform_params(self):
        p = np.empty(self.num_isoform_params, np.float32)
        check(L.lib().polee_regression_get_isoform_params(self._h, ptr(p, f32p)), self.ctx._h)
        return p

    def set_isoform_params(self, p):
        p = arr(p, np.float32).reshape(-1)
        if p.size != self.num_isoform_params:
            raise ValueError("expected %d isoform parameters" % self.num_isoform_params)
        check(L.lib().polee_regression_set_isoform_params(self._h, ptr(p, f32p)), self.ctx._h)

    def isoform_gradients(self):
        """Gradient of the isoform block left by the last loss_and_gradients()."""
        g = np.empty(self.num_isoform_params, np.float32)
        check(L.lib().polee_regression_get_isoform_grad(self._h, ptr(g, f32p)), self.ctx._h)
        return g

    # test hooks: the two halves of a step (include/polee_hip_debug.h)
    def _data_pass(self, noise):
        z = arr(noise, np.float32).reshape(-1)
        f = L.lib().polee_debug_regression_num_stats
        f.restype, f.argtypes = C.c_int64, [C.c_void_p]
        stats = np.empty(int(f(self._h)), np.float32)
        check(L.lib().polee_debug_regression_data_pass(self._h, ptr(z, f32p), ptr(stats, f32p)), self.ctx._h)
        return stats

    def _prior_pass(self, stats):
        st = arr(stats, np.float32).reshape(-1)
        loss, g = np.empty(1, np.float32), np.empty(self.num_params, np.float32)
        check(L.lib().polee_debug_regression_prior_pass(self._h, ptr(st, f32p), ptr(loss, f32p), ptr(g, f32p)),
              self.ctx._h)
        return float(loss[0]), g

    # ---- classify (models/polee_regression.py:342-413)
    def set_design(self, F):
        Fm = arr(np.atleast_2d(F), np.float32)
        if Fm.shape != (self.num_samples, self.num_factors):
            raise ValueError("the design matrix must be [S, F]")
        check(L.lib().polee_regression_set_design(self._h, ptr(Fm, f32p)), self.ctx._h)
        self.design = Fm

    def design_gradient(self):
        """d loss / d design [S, F] of the last evaluation or fit step (after set_design)."""
        g = np.empty((self.num_samples, self.num_factors), np.float32)
        check(L.lib().polee_regression_design_grad(self._h, ptr(g, f32p)), self.ctx._h)
        return g

    def _shared_size(self):
        """flat parameters in front of qx_loc: everything the samples share (the layout of include/polee_hip.h)"""
        F, n, d = self.num_factors, self.num_features, self.kernel_regression_degree
        return 4 + F * d + 2 * d + 10 * F * n + 4 * n

    def classify(self, x_init, likelihood_model, surrogate_likelihood_model, sample_scales, use_point_estimates, niter,
                 extra_training_vars=(), seed=123456789, return_trace=False):
        """classify (models/polee_regression.py:342-413): class probabilities [S_test, F] of testing samples under the FITTED model.
        Their design matrix is latent -- F ~ OneHotCategorical(uniform) per sample, surrogate RelaxedOneHotCategorical(T, logits)
        with the temperature annealed from 5 to 0.5 over the run (:385-391) -- everything the fitted model shares keeps its
        surrogate, and the trainable variables are the logits (+ the testing samples' qx_loc / qx_softplus_scale unless
        use_point_estimates), Adam at 1e-3 (:400).  `likelihood_model`: the testing samples' RNASeqApproxLikelihood (None with point
        estimates); `surrogate_likelihood_model` is part of the reference's signature only.  Per step the device model over the testing
        samples draws every latent but F, evaluates the loss and the gradients (polee_regression_fit, one step) and returns d loss / d F;
        the relaxed rows, their density and the logits' Adam are a few dozen numbers per step and stay on the host.
        Returns softmax(logits) (:407), with return_trace also the loss trace."""
        if extra_training_vars or isinstance(self, (RNASeqGeneLinearRegression, RNASeqGeneIsoformLinearRegression, RNASeqJointLinearRegression,
                                                    RNASeqNormalTranscriptLinearRegression)):
            raise NotImplementedError("classify is built for the transcript-level model (the one models/imputation.jl uses)")
        x0 = arr(np.atleast_2d(x_init), np.float32)
        S, K, n = x0.shape[0], self.num_factors, self.num_features
        c = self._ctor
        test = RNASeqLinearRegression(
            np.full((S, K), 1.0 / K, np.float32), x0, None if use_point_estimates else likelihood_model, c["x_bias_loc0"],
            c["x_bias_scale0"], c["x_scale_hinges"] if c["x_scale_hinges"] is not None else self._default_hinges(),
            sample_scales, c["use_distortion"], c["scale_penalty"], use_point_estimates, self.kernel_regression_degree,
            c["kernel_regression_bandwidth"], ctx=self.ctx, x_init_mean=c["x_init_mean"])
        ns = self._shared_size()
        flat = np.concatenate([self.get_flat_params()[:ns], x0.reshape(-1), np.full(S * n, -1.0, np.float32)]).astype(np.float32)
        test.set_flat_params(flat)
        lib = L.lib()
        check(lib.polee_regression_set_learning_rate(test._h, C.c_float(1e-3)), self.ctx._h)
        check(lib.polee_regression_set_trainable(test._h, C.c_int64(ns if not use_point_estimates else test.num_params),
                                                 C.c_int64(test.num_params)), self.ctx._h)
        rng = np.random.default_rng(seed)
        logits = np.zeros((S, K))
        m, v = np.zeros_like(logits), np.zeros_like(logits)
        trace = np.empty(int(niter), np.float64)
        for step in range(1, int(niter) + 1):
            T = 5.0 if step == 1 else 5.0 * 0.1 ** (step / float(niter))  # (trace_fn :385-391 anneals AFTER a step)
            gum = -np.log(-np.log(rng.uniform(1e-12, 1.0, size=(S, K))))
            y = _softmax((logits + gum) / T)
            test.set_design(y)
            dev_loss = test._fit_steps(1, seed + step)[0]
            lq, dl_direct, dl_dy = relaxed_onehot_terms(logits, y, T)
            dy = dl_dy + test.design_gradient().astype(np.float64)
            grad = dl_direct + (y * (dy - (y * dy).sum(axis=1, keepdims=True))) / T  # through y = softmax((logits + g) / T)
            trace[step - 1] = float(dev_loss) + lq.sum() + S * math.log(K)  # - log p(F) = log K per sample (uniform prior, :349-351)
            m = 0.9 * m + 0.1 * grad
            v = 0.999 * v + 0.001 * grad * grad
            lr_t = 1e-3 * math.sqrt(1.0 - 0.999 ** step) / (1.0 - 0.9 ** step)
            logits -= lr_t * m / (np.sqrt(v) + 1e-7)
        probs = _softmax(logits)
        return (probs, trace) if return_trace else probs

    def _fit_steps(self, niter, seed):
        """polee_regression_fit without the download of every variable that fit() returns: the loss trace only"""
        trace = np.empty(int(niter), np.float32)
        check(L.lib().polee_regression_fit(self._h, int(niter), C.c_uint64(seed), None, ptr(trace, f32p)), self.ctx._h)
        return trace

    def _default_hinges(self):
        """choose_knots (src/polee.py:69-76) over the training samples' column means, as polee_regression_create derives them"""
        mean, d = self._ctor["x_init_mean"], self.kernel_regression_degree
        lo, hi = float(mean.min()), float(mean.max())
        return np.array([lo + (k + 1) * (hi - lo) / (d + 1) for k in range(d)], np.float32)

    def fit(self, niter, seed=123456789, noise=None, return_trace=False):
        """fit (models/polee_regression.py:303-340): returns (qx_loc, qw_loc, qw_scale, qx_bias_loc, qx_scale)."""
        z = None if noise is None else arr(noise, np.float32).reshape(-1)
        if z is not None and z.size != int(niter) * self.num_noise:
            raise ValueError("noise must hold niter x num_noise values")
        trace = np.empty(int(niter), np.float32)
        check(L.lib().polee_regression_fit(self._h, int(niter), C.c_uint64(seed), ptr(z, f32p), ptr(trace, f32p)),
              self.ctx._h)
        v = self.variables()
        out = (v["qx_loc"], v["qw_loc"], _softplus(v["qw_softplus_scale"]), v["qx_bias_loc"],
               _softplus(v["qx_scale_loc"]))
        return out + (trace,) if return_trace else out


class RNASeqTranscriptLinearRegression(RNASeqLinearRegression):
    """RNASeqTranscriptLinearRegression (models/polee_regression.py:422-460).  `vars`: the dict of
    create_tensorflow_variables! (estimate.jl:502-556) or an RNASeqApproxLikelihood."""

    def __init__(self, vars, x_init, F_arr, sample_scales, use_distortion, scale_penalty, use_point_estimates,
                 kernel_regression_degree=15, kernel_regression_bandwidth=1.0, ctx=None, comm=None, x_init_mean=None,
                 x_scale_hinges=None):
        x_init = np.asarray(x_init, np.float32)
        num_features = x_init.shape[1]
        lik = None
        if not use_point_estimates:
            lik = vars if isinstance(vars, RNASeqApproxLikelihood) else RNASeqApproxLikelihood(vars, ctx=ctx)
        super().__init__(F_arr, x_init, lik, math.log(1.0 / num_features), 12.0, x_scale_hinges, sample_scales,
                         use_distortion, scale_penalty, use_point_estimates, kernel_regression_degree,
                         kernel_regression_bandwidth, ctx=ctx, comm=comm, x_init_mean=x_init_mean)


    def classify(self, vars, x_init, sample_scales, use_point_estimates, niter, seed=123456789, return_trace=False):
        """classify (models/polee_regression.py:462-483, as models/imputation.jl:208-213 calls it): `vars` are the TESTING samples'
        likelihood variables (or their RNASeqApproxLikelihood); unused with point estimates."""
        lik = None
        if not use_point_estimates:
            lik = vars if isinstance(vars, RNASeqApproxLikelihood) else RNASeqApproxLikelihood(vars, ctx=self.ctx)
        return super().classify(x_init, lik, None, sample_scales, use_point_estimates, niter, seed=seed, return_trace=return_trace)


class RNASeqGeneLinearRegression(RNASeqLinearRegression):
    """RNASeqGeneLinearRegression (models/polee_regression.py:533-600): regression over gene expression with the
    transcript-level approximate likelihood reached through within-gene isoform log-expression.
    feature_idxs / transcript_idxs: the 1-based (gene, transcript) pairs of the reference; feature_sizes is unused
    (the reference only forwards it).  Built without point estimates only."""

    def __init__(self, vars, feature_idxs, transcript_idxs, x_gene_init, x_isoform_init, feature_sizes, F_arr,
                 sample_scales, use_distortion, scale_penalty, use_point_estimates, kernel_regression_degree=15,
                 kernel_regression_bandwidth=1.0, ctx=None):
        if use_point_estimates:
            raise NotImplementedError("the gene-level model is built without point estimates only")
        x_gene_init = np.asarray(x_gene_init, np.float32)
        lik = vars if isinstance(vars, RNASeqApproxLikelihood) else RNASeqApproxLikelihood(vars, ctx=ctx)
        fi = np.asarray(feature_idxs, np.int64).reshape(-1) - 1
        ti = np.asarray(transcript_idxs, np.int64).reshape(-1) - 1
        gene_of = np.full(lik.n, -1, np.int64)
        gene_of[ti] = fi
        if (gene_of < 0).any():
            raise ValueError("every transcript must belong to a gene")
        super().__init__(F_arr, x_gene_init, None, math.log(1.0 / x_gene_init.shape[1]), 12.0, None, sample_scales,
                         use_distortion, scale_penalty, False, kernel_regression_degree, kernel_regression_bandwidth,
                         ctx=ctx or lik.ctx, gene_likelihood=(lik, gene_of, x_isoform_init))

    def isoform_variables(self):
        v, S, nt = self.get_isoform_params(), self.num_samples, self.likelihood_model.n
        return dict(qx_isoform_mean_loc=v[:nt], qx_isoform_mean_softplus_scale=v[nt:2 * nt],
                    qx_isoform_loc=v[2 * nt:2 * nt + S * nt].reshape(S, nt),
                    qx_isoform_softplus_scale=v[2 * nt + S * nt:].reshape(S, nt))


class RNASeqGeneIsoformLinearRegression(RNASeqLinearRegression):
    """RNASeqGeneIsoformLinearRegression (models/polee_regression.py:656-877): regression over gene expression AND over
    the within-gene isoform mixtures (own design matrix F_isoform_arr, horseshoe+ coefficients).  Argument order and
    fit()'s return follow the reference.  Built without point estimates only."""

    # the isoform block in the order of the flat vector (include/polee_hip.h); shapes: 1, [Fi, nt], [nt], [S, nt]
    ISOFORM_PARAMS = [
        ("qw_isoform_global_scale_variance_loc", "1"), ("qw_isoform_global_scale_variance_softplus_scale", "1"),
        ("qw_isoform_global_scale_noncentered_loc", "1"), ("qw_isoform_global_scale_noncentered_softplus_scale", "1"),
        ("qw_isoform_local1_scale_variance_loc", "Ft"), ("qw_isoform_local1_scale_variance_softplus_scale", "Ft"),
        ("qw_isoform_local1_scale_noncentered_loc", "Ft"), ("qw_isoform_local1_scale_noncentered_softplus_scale", "Ft"),
        ("qw_isoform_local2_scale_variance_loc", "Ft"), ("qw_isoform_local2_scale_variance_softplus_scale", "Ft"),
        ("qw_isoform_local2_scale_noncentered_loc", "Ft"), ("qw_isoform_local2_scale_noncentered_softplus_scale", "Ft"),
        ("qw_isoform_loc", "Ft"), ("qw_isoform_softplus_scale", "Ft"),
        ("qx_isoform_bias_loc", "t"), ("qx_isoform_bias_softplus_scale", "t"),
        ("qx_isoform_scale_loc", "t"), ("qx_isoform_scale_softplus_scale", "t"),
        ("qx_isoform_loc", "St"), ("qx_isoform_softplus_scale", "St"),
    ]

    def __init__(self, vars, feature_idxs, transcript_idxs, x_gene_init, x_isoform_init, feature_sizes, F_gene_arr,
                 F_isoform_arr, sample_scales, use_distortion, scale_penalty, use_point_estimates,
                 kernel_regression_degree=15, kernel_regression_bandwidth=1.0, ctx=None):
        if use_point_estimates:
            raise NotImplementedError("the gene-isoform model is built without point estimates only")
        x_gene_init = np.asarray(x_gene_init, np.float32)
        lik = vars if isinstance(vars, RNASeqApproxLikelihood) else RNASeqApproxLikelihood(vars, ctx=ctx)
        fi = np.asarray(feature_idxs, np.int64).reshape(-1) - 1
        ti = np.asarray(transcript_idxs, np.int64).reshape(-1) - 1
        gene_of = np.full(lik.n, -1, np.int64)
        gene_of[ti] = fi
        if (gene_of < 0).any():
            raise ValueError("every transcript must belong to a gene")
        super().__init__(F_gene_arr, x_gene_init, None, math.log(1.0 / x_gene_init.shape[1]), 12.0, None, sample_scales,
                         use_distortion, scale_penalty, False, kernel_regression_degree, kernel_regression_bandwidth,
                         ctx=ctx or lik.ctx, gene_likelihood=(lik, gene_of, x_isoform_init, F_isoform_arr))

    def isoform_variables(self):
        v, S, nt, Fi = self.get_isoform_params(), self.num_samples, self.likelihood_model.n, self.num_isoform_factors
        shapes = {"1": (), "Ft": (Fi, nt), "t": (nt,), "St": (S, nt)}
        out, o = {}, 0
        for name, code in self.ISOFORM_PARAMS:
            k = int(np.prod(shapes[code], dtype=np.int64))
            out[name] = v[o:o + k].reshape(shapes[code])
            o += k
        assert o == v.size
        return out

    def fit(self, niter, seed=123456789, noise=None, return_trace=False):
        """fit (models/polee_regression.py:833-857): (qw_gene_loc, qw_gene_scale, qw_isoform_loc, qw_isoform_scale,
        qx_isoform_bias_loc, qx_isoform_bias_scale, qx_isoform_scale, qx_gene_bias_loc, qx_gene_scale,
        qx_gene_loc_factor_est)."""
        base = super().fit(niter, seed=seed, noise=noise, return_trace=True)
        _, qw_gene_loc, qw_gene_scale, qx_gene_bias_loc, qx_gene_scale, trace = base
        iv = self.isoform_variables()
        out = (qw_gene_loc, qw_gene_scale, iv["qw_isoform_loc"], _softplus(iv["qw_isoform_softplus_scale"]),
               iv["qx_isoform_bias_loc"], _softplus(iv["qx_isoform_bias_softplus_scale"]),
               _softplus(iv["qx_isoform_scale_loc"]), qx_gene_bias_loc, qx_gene_scale,
               self.design @ qw_gene_loc)
        return out + (trace,) if return_trace else out

    def write_other_params(self, output_filename):
        """write_other_params (models/polee_regression.py:859-875)"""
        iv = self.isoform_variables()
        with open(output_filename, "w") as output:
            for name in ("global_scale_variance_loc", "global_scale_variance_softplus_scale",
                         "global_scale_noncentered_loc", "global_scale_noncentered_softplus_scale"):
                output.write("qw_isoform_{}_var: {}\n".format(name, iv["qw_isoform_" + name]))


class RNASeqJointLinearRegression(RNASeqLinearRegression):
    """RNASeqJointLinearRegression (models/polee_regression.py:879-1283, driven by models/joint-regression.jl): regression
    over gene (TSS-group) expression AND over splice-feature usage, whose predictor reaches the transcripts through the 0/1
    feature matrix.  Argument order follows the reference: tss_is / tss_js = 1-based (transcript, gene) pairs,
    feature_is / feature_js = 1-based (transcript, splice feature) pairs.  fit() returns (qw_gene_loc, qw_gene_scale,
    qw_splice_loc, qw_splice_scale) (:1230-1234).  Built without point estimates only."""

    # the splice block of the isoform-parameter vector (include/polee_hip.h); shapes: 1, [F, P], [P]; then [nt], [S, nt]
    SPLICE_PARAMS = [
        ("qw_splice_global_scale_variance_loc", "1"), ("qw_splice_global_scale_variance_softplus_scale", "1"),
        ("qw_splice_global_scale_noncentered_loc", "1"), ("qw_splice_global_scale_noncentered_softplus_scale", "1"),
        ("qw_splice_local_scale_variance_loc", "FP"), ("qw_splice_local_scale_variance_softplus_scale", "FP"),
        ("qw_splice_local_scale_noncentered_loc", "FP"), ("qw_splice_local_scale_noncentered_softplus_scale", "FP"),
        ("_unused_local2_a", "FP"), ("_unused_local2_b", "FP"), ("_unused_local2_c", "FP"), ("_unused_local2_d", "FP"),
        ("qw_splice_loc", "FP"), ("qw_splice_softplus_scale", "FP"),
        ("qx_splice_bias_loc", "P"), ("qx_splice_bias_softplus_scale", "P"),
        ("_unused_scale_a", "P"), ("_unused_scale_b", "P"),
        ("qx_iso_scale_loc", "t"), ("qx_iso_scale_softplus_scale", "t"),
        ("qx_iso_loc", "St"), ("qx_iso_softplus_scale", "St"),
    ]

    def __init__(self, vars, tss_is, tss_js, num_gene_features, feature_is, feature_js, num_splice_features, gene_sizes,
                 x_gene_init, x_isoform_init, F_arr, sample_scales, use_point_estimates, kernel_regression_degree=15,
                 kernel_regression_bandwidth=1.0, ctx=None):
        if use_point_estimates:
            raise NotImplementedError("the joint model is built without point estimates only")
        x_gene_init = np.asarray(x_gene_init, np.float32)
        if x_gene_init.shape[1] != int(num_gene_features):
            raise ValueError("x_gene_init must be [S, num_gene_features]")
        lik = vars if isinstance(vars, RNASeqApproxLikelihood) else RNASeqApproxLikelihood(vars, ctx=ctx)
        ti = np.asarray(tss_is, np.int64).reshape(-1) - 1
        gi = np.asarray(tss_js, np.int64).reshape(-1) - 1
        gene_of = np.full(lik.n, -1, np.int64)
        gene_of[ti] = gi
        if (gene_of < 0).any():
            raise ValueError("every transcript must belong to a gene feature")
        # the gene block: no distortion (:1029), scale-drift penalty Normal(0, 5e-4) (:1052-1054)
        super().__init__(F_arr, x_gene_init, None, math.log(1.0 / int(num_gene_features)), 12.0, None, sample_scales, False,
                         5e-4, False, kernel_regression_degree, kernel_regression_bandwidth, ctx=ctx or lik.ctx,
                         gene_likelihood=(lik, gene_of, x_isoform_init))
        # (the base constructor attaches the plain gene-level likelihood; polee_regression_set_joint_likelihood below
        # re-attaches it and replaces its isoform block by the joint model's splice block)
        xi0 = arr(np.atleast_2d(x_isoform_init), np.float32)
        if xi0.shape != (self.num_samples, lik.n):
            raise ValueError("x_isoform_init must be [S, nt]")
        pt = arr(np.asarray(feature_is, np.int64).reshape(-1) - 1, np.int32)
        pf = arr(np.asarray(feature_js, np.int64).reshape(-1) - 1, np.int32)
        if pt.size != pf.size:
            raise ValueError("feature_is and feature_js must pair up")
        self.num_splice_features = int(num_splice_features)
        self.likelihood_model = lik
        lib = L.lib()
        check(lib.polee_regression_set_joint_likelihood(self._h, lik._h, ptr(arr(gene_of, np.int32), L.i32p), ptr(xi0, f32p),
                                                        self.num_splice_features, ptr(pt, L.i32p), ptr(pf, L.i32p),
                                                        C.c_int64(pt.size)), self.ctx._h)
        lib.polee_regression_num_isoform_params.restype = C.c_int64
        lib.polee_regression_num_isoform_params.argtypes = [C.c_void_p]
        self.num_isoform_params = int(lib.polee_regression_num_isoform_params(self._h))
        self.num_noise = int(lib.polee_regression_num_noise(self._h))

    def splice_variables(self):
        v, S, nt, F, P = (self.get_isoform_params(), self.num_samples, self.likelihood_model.n, self.num_factors,
                          self.num_splice_features)
        shapes = {"1": (), "FP": (F, P), "P": (P,), "t": (nt,), "St": (S, nt)}
        out, o = {}, 0
        for name, code in self.SPLICE_PARAMS:
            k = int(np.prod(shapes[code], dtype=np.int64))
            out[name] = v[o:o + k].reshape(shapes[code])
            o += k
        assert o == v.size
        return out

    def fit(self, niter, seed=123456789, noise=None, return_trace=False):
        """fit (models/polee_regression.py:1204-1234): (qw_gene_loc, qw_gene_scale, qw_splice_loc, qw_splice_scale)."""
        base = super().fit(niter, seed=seed, noise=noise, return_trace=True)
        _, qw_gene_loc, qw_gene_scale, _, _, trace = base
        sv = self.splice_variables()
        out = (qw_gene_loc, qw_gene_scale, sv["qw_splice_loc"], _softplus(sv["qw_splice_softplus_scale"]))
        return out + (trace,) if return_trace else out


class RNASeqNormalTranscriptLinearRegression(RNASeqLinearRegression):
    """RNASeqNormalTranscriptLinearRegression (models/polee_regression.py:490-531): point estimates and their standard
    deviation in place of the approximate likelihood.  `vars` is unused, as in the reference."""

    def __init__(self, vars, x_likelihood_loc, x_likelihood_scale, F_arr, sample_scales, use_distortion, scale_penalty,
                 kernel_regression_degree=15, kernel_regression_bandwidth=1.0, ctx=None, comm=None, x_init_mean=None):
        loc = np.asarray(x_likelihood_loc, np.float32)
        super().__init__(F_arr, loc, None, math.log(1.0 / loc.shape[1]), 12.0, None, sample_scales, use_distortion,
                         scale_penalty, False, kernel_regression_degree, kernel_regression_bandwidth, ctx=ctx, comm=comm,
                         x_init_mean=x_init_mean, normal_likelihood=(loc, x_likelihood_scale))


# ---- output semantics (src/regression.jl:604-685)
def find_minimum_effect_size(mu, sigma, target_coverage):
    """Bisection of src/regression.jl:604-622 on P(|w| < delta) under Normal(mu, sigma)."""
    from scipy.stats import norm
    lo, hi, coverage = 0.0, 20.0, 1.0
    while abs(coverage - target_coverage) / target_coverage > 0.001:
        d = (hi + lo) / 2
        coverage = norm.cdf(d, mu, sigma) - norm.cdf(-d, mu, sigma)
        if coverage > target_coverage:
            hi = d
        else:
            lo = d
        if hi - lo < 1e-15:
            break
    return (hi + lo) / 2


def write_regression_effects(output_filename, factor_names, feature_names_label, feature_names, qx_bias, qx_scale,
                             qw_loc, qw_scale, q0, q1, effect_size, mes_target_coverage,
                             write_variational_posterior_params=False):
    """write_regression_effects (src/regression.jl:625-685): CSV of effect sizes in log2 units with t_10 credible
    intervals and the minimum effect size."""
    from scipy.stats import t as tdist
    qw_loc, qw_scale = np.asarray(qw_loc), np.asarray(qw_scale)
    assert qw_loc.shape == qw_scale.shape
    num_factors, num_features = qw_loc.shape
    ln2 = math.log(2.0)
    tq0, tq1 = tdist.ppf(q0, 10.0), tdist.ppf(q1, 10.0)
    es = None if effect_size is None else math.log(abs(effect_size))
    with open(output_filename, "w") as out:
        out.write("factor,%s,min_effect_size,mean_effect_size,lower_credible,upper_credible" % feature_names_label)
        if es is not None:
            out.write(",prob_de,prob_down_de,prob_up_de")
        if write_variational_posterior_params:
            out.write(",qx_bias_loc,qx_scale,qw_loc,qw_scale")
        out.write("\n")
        for i in range(num_factors):
            for j in range(num_features):
                loc, sc = float(qw_loc[i, j]), float(qw_scale[i, j])
                mes = find_minimum_effect_size(loc, sc, mes_target_coverage)
                out.write("%s,%s,%f,%f,%f,%f" % (factor_names[i], feature_names[j], mes / ln2, loc / ln2,
                                                 (tq0 * sc + loc) / ln2, (tq1 * sc + loc) / ln2))
                if es is not None:
                    down = tdist.cdf((-es - loc) / sc, 10.0)
                    up = tdist.sf((es - loc) / sc, 10.0)
                    out.write(",%f,%f,%f" % (max(down, up), down, up))
                if write_variational_posterior_params:
                    out.write(",%f,%f,%f,%f" % (qx_bias[j], qx_scale[j], loc, sc))
                out.write("\n")
