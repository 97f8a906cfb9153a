"""TEST INFRASTRUCTURE ONLY -- CPU (NumPy, float64) restatement of the regression model's variational loss.

Follows `models/polee_regression.py:18-340` (RNASeqLinearRegression.model_fn / variational_model_fn / fit) and the
helpers in `src/polee.py:14-76`.  The loss is tfp.vi.fit_surrogate_posterior's reverse-KL Monte-Carlo estimate at
sample_size = 1:   loss = log q(z) - log p(z),  z one reparameterised draw of the surrogate posterior.

PARITY UNPINNED: TensorFlow / TensorFlow-Probability are not available in this image and the reference pins no
version of either (README.md:74 asks only for "the tensorflow python library").  The log-densities below are TFP's
published definitions:
  Normal(l, s).log_prob(x)        = -0.5 ((x-l)/s)^2 - log s - 0.5 log 2pi
  HalfNormal(s).log_prob(x)       = 0.5 log(2/pi) - log s - 0.5 (x/s)^2
  InverseGamma(a, b).log_prob(x)  = a log b - lgamma(a) - (a+1) log x - b/x
  Cauchy(l, s).log_prob(x)        = -log(pi s) - log1p(((x-l)/s)^2)
  HalfCauchy(0, s).log_prob(x)    = log 2 - log(pi s) - log1p((x/s)^2)
  Deterministic.log_prob          = 0 at its own sample
  TransformedDistribution(Normal, Softplus).log_prob(y) = Normal.log_prob(u) - log sigmoid(u),  y = softplus(u)
Only tests may import this module.
"""
import numpy as np
from scipy.special import gammaln

LOG2PI = np.log(2.0 * np.pi)

# (name, shape-code) in the order of the flat parameter vector shared with the device code
# shape codes: '1' scalar, 'Fd' [F,deg], 'd' [deg], 'Fn' [F,n], 'n' [n], 'Sn' [S,n]
PARAMS = [
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
# noise of the reparameterised draws, same convention
NOISE = [
    ("w_global_scale_variance", "1"), ("w_global_scale_noncentered", "1"),
    ("w_local1_scale_variance", "Fn"), ("w_local1_scale_noncentered", "Fn"),
    ("w_local2_scale_variance", "Fn"), ("w_local2_scale_noncentered", "Fn"),
    ("w", "Fn"), ("x_bias", "n"), ("x_scale", "n"), ("x", "Sn"),
]


def shape_of(code, S, F, n, deg):
    return {"1": (1,), "Fd": (F, deg), "d": (deg,), "Fn": (F, n), "n": (n,), "Sn": (S, n)}[code]


def unflatten(vec, table, S, F, n, deg):
    out, o = {}, 0
    for name, code in table:
        shp = shape_of(code, S, F, n, deg)
        k = int(np.prod(shp))
        out[name] = np.asarray(vec[o:o + k], np.float64).reshape(shp)
        o += k
    assert o == len(vec)
    return out


def flatten(d, table):
    return np.concatenate([np.asarray(d[name], np.float64).reshape(-1) for name, _ in table])


def softplus(x):
    return np.logaddexp(0.0, x)


def log_sigmoid(x):
    return -np.logaddexp(0.0, -x)


def choose_knots(low, high, degree):
    """src/polee.py:69-76"""
    d = (high - low) / (degree + 1)
    return np.array([low + (i + 1) * d for i in range(degree)])


def kernel_regression_weights(bandwidth, mean, hinges):
    """src/polee.py:36-47 -> [deg, n]"""
    diffs = mean[None, :] - hinges[:, None]
    w = np.clip(np.exp(-np.square(diffs / bandwidth)), 1e-10, 1.0)
    return w / w.sum(axis=0, keepdims=True)


def initial_params(x_init, F, deg):
    """RNASeqLinearRegression.__init__ (models/polee_regression.py:49-119)"""
    S, n = x_init.shape
    p = {}
    for name, code in PARAMS:
        shp = shape_of(code, S, F, n, deg)
        if name.endswith("softplus_scale"):
            p[name] = np.full(shp, -1.0)
        else:
            p[name] = np.zeros(shp)
    p["qw_softplus_scale"][:] = 0.0
    p["qx_bias_loc"] = x_init.mean(axis=0).astype(np.float64)
    p["qx_scale_concentration_c_loc"][:] = 1.0
    p["qx_scale_scale_c_loc"][:] = 1.0
    p["qx_scale_loc"][:] = -0.5
    p["qx_loc"] = x_init.astype(np.float64).copy()
    return p


def _normal_lp(x, loc, scale):
    return -0.5 * np.square((x - loc) / scale) - np.log(scale) - 0.5 * LOG2PI


def _invgamma_lp(x, a, b):
    return a * np.log(b) - gammaln(a) - (a + 1.0) * np.log(x) - b / x


def _halfnormal_lp(x):
    return 0.5 * np.log(2.0 / np.pi) - 0.5 * np.square(x)


ISO_PARAMS = [("qx_isoform_mean_loc", "t"), ("qx_isoform_mean_softplus_scale", "t"), ("qx_isoform_loc", "St"),
              ("qx_isoform_softplus_scale", "St")]
ISO_NOISE = [("x_isoform_mean", "t"), ("x_isoform", "St")]


def unflatten_iso(vec, table, S, nt):
    out, o = {}, 0
    for name, code in table:
        shp = (nt,) if code == "t" else (S, nt)
        k = int(np.prod(shp))
        out[name] = np.asarray(vec[o:o + k], np.float64).reshape(shp)
        o += k
    assert o == len(vec)
    return out


def isoform_terms(ip, ie, x_gene, gene_lik=None):
    """Gene-level likelihood model (RNASeqGeneLinearRegression.likelihood_model / surrogate_likelihood_model,
    models/polee_regression.py:558-596): returns log q - log p of the isoform block (with -gene_lik(x_gene, x_isoform)
    when given) and the draws."""
    sm = softplus(ip["qx_isoform_mean_softplus_scale"])
    mean = ip["qx_isoform_mean_loc"] + sm * ie["x_isoform_mean"]
    sx = softplus(ip["qx_isoform_softplus_scale"])
    xi = ip["qx_isoform_loc"] + sx * ie["x_isoform"]
    logq = np.sum(_normal_lp(mean, ip["qx_isoform_mean_loc"], sm)) + np.sum(_normal_lp(xi, ip["qx_isoform_loc"], sx))
    logp = np.sum(_normal_lp(mean, 0.0, 2.0)) + np.sum(_normal_lp(xi, mean[None, :], 1.0))
    if gene_lik is not None:
        logp += float(np.sum(gene_lik(x_gene, xi)))
    return float(logq - logp), xi


def isoform_regression_terms(ip, ie, design_isoform, x_gene, gene_lik=None):
    """Gene-isoform model (RNASeqGeneIsoformLinearRegression.likelihood_model / surrogate_likelihood_model,
    models/polee_regression.py:696-733 and :777-830): log q - log p of the isoform block at the draw `ie` (with
    -gene_lik(x_gene, x_isoform) when given) and the draws.  `ip` / `ie`: PARAMS / NOISE unflattened with F = the
    isoform factors, n = the transcripts, deg = 0 (the block has the model's own order without the hinge arrays)."""
    logq = 0.0

    def sp_normal(loc, sraw, e):
        nonlocal logq
        s = softplus(ip[sraw])
        u = ip[loc] + s * e
        logq += np.sum(_normal_lp(u, ip[loc], s) - log_sigmoid(u))
        return softplus(u)

    def normal(loc, sraw, e):
        nonlocal logq
        s = softplus(ip[sraw])
        v = ip[loc] + s * e
        logq += np.sum(_normal_lp(v, ip[loc], s))
        return v

    gv = sp_normal("qw_global_scale_variance_loc", "qw_global_scale_variance_softplus_scale", ie["w_global_scale_variance"])
    gn = sp_normal("qw_global_scale_noncentered_loc", "qw_global_scale_noncentered_softplus_scale",
                   ie["w_global_scale_noncentered"])
    l1v = sp_normal("qw_local1_scale_variance_loc", "qw_local1_scale_variance_softplus_scale", ie["w_local1_scale_variance"])
    l1n = sp_normal("qw_local1_scale_noncentered_loc", "qw_local1_scale_noncentered_softplus_scale",
                    ie["w_local1_scale_noncentered"])
    l2v = sp_normal("qw_local2_scale_variance_loc", "qw_local2_scale_variance_softplus_scale", ie["w_local2_scale_variance"])
    l2n = sp_normal("qw_local2_scale_noncentered_loc", "qw_local2_scale_noncentered_softplus_scale",
                    ie["w_local2_scale_noncentered"])
    w = normal("qw_loc", "qw_softplus_scale", ie["w"])
    bias = normal("qx_bias_loc", "qx_bias_softplus_scale", ie["x_bias"])
    xscale = sp_normal("qx_scale_loc", "qx_scale_softplus_scale", ie["x_scale"])
    xi = normal("qx_loc", "qx_softplus_scale", ie["x"])
    logp = np.sum(_invgamma_lp(gv, 0.5, 0.5)) + np.sum(_halfnormal_lp(gn))
    logp += np.sum(_invgamma_lp(l1v, 0.5, 0.5)) + np.sum(_halfnormal_lp(l1n))
    logp += np.sum(_invgamma_lp(l2v, 0.5, 0.5)) + np.sum(_halfnormal_lp(l2n))
    w_scale = l1n * np.sqrt(l1v) * (l2n * np.sqrt(l2v)) * (gn * np.sqrt(gv))
    logp += np.sum(_normal_lp(w, 0.0, w_scale))
    logp += np.sum(_normal_lp(bias, 0.0, 2.0))
    loc = bias[None, :] + np.asarray(design_isoform, np.float64) @ w
    logp += np.sum(_invgamma_lp(xscale, 0.001, 0.001))
    logp += np.sum(_normal_lp(xi, loc, xscale[None, :]))
    if gene_lik is not None:
        logp += float(np.sum(gene_lik(x_gene, xi)))
    return float(logq - logp), xi


def regression_loss(p, eps, design, W, sample_scales, x_bias_loc0, x_bias_scale0, use_distortion, scale_penalty,
                    use_point_estimates, lik=None):
    """loss = log q - log p at the draw defined by `eps`.  `lik(x) -> lp [S]` is the approximate likelihood
    (polee_approx_likelihood.py:367-450); ignored with point estimates.  Returns (loss, draws)."""
    logq = 0.0
    z = {}

    def sp_normal(name_loc, name_s, e):
        nonlocal logq
        s = softplus(p[name_s])
        u = p[name_loc] + s * e
        logq += np.sum(_normal_lp(u, p[name_loc], s) - log_sigmoid(u))
        return softplus(u)

    def normal(name_loc, name_s, e):
        nonlocal logq
        s = softplus(p[name_s])
        v = p[name_loc] + s * e
        logq += np.sum(_normal_lp(v, p[name_loc], s))
        return v

    gv = sp_normal("qw_global_scale_variance_loc", "qw_global_scale_variance_softplus_scale",
                   eps["w_global_scale_variance"])
    gn = sp_normal("qw_global_scale_noncentered_loc", "qw_global_scale_noncentered_softplus_scale",
                   eps["w_global_scale_noncentered"])
    l1v = sp_normal("qw_local1_scale_variance_loc", "qw_local1_scale_variance_softplus_scale",
                    eps["w_local1_scale_variance"])
    l1n = sp_normal("qw_local1_scale_noncentered_loc", "qw_local1_scale_noncentered_softplus_scale",
                    eps["w_local1_scale_noncentered"])
    l2v = sp_normal("qw_local2_scale_variance_loc", "qw_local2_scale_variance_softplus_scale",
                    eps["w_local2_scale_variance"])
    l2n = sp_normal("qw_local2_scale_noncentered_loc", "qw_local2_scale_noncentered_softplus_scale",
                    eps["w_local2_scale_noncentered"])
    w = normal("qw_loc", "qw_softplus_scale", eps["w"])
    x_bias = normal("qx_bias_loc", "qx_bias_softplus_scale", eps["x_bias"])
    dist_c = p["qw_distortion_c_loc"]
    conc_c = softplus(p["qx_scale_concentration_c_loc"])
    scale_c = softplus(p["qx_scale_scale_c_loc"])
    x_scale = sp_normal("qx_scale_loc", "qx_scale_softplus_scale", eps["x_scale"])
    if use_point_estimates:
        x = p["qx_loc"]
    else:
        x = normal("qx_loc", "qx_softplus_scale", eps["x"])

    # ---- log p (model_fn, models/polee_regression.py:124-256)
    logp = 0.0
    logp += np.sum(_invgamma_lp(gv, 0.5, 0.5)) + np.sum(_halfnormal_lp(gn))
    logp += np.sum(_invgamma_lp(l1v, 0.5, 0.5)) + np.sum(_halfnormal_lp(l1n))
    logp += np.sum(_invgamma_lp(l2v, 0.5, 0.5)) + np.sum(_halfnormal_lp(l2n))
    w_scale = (l1n * np.sqrt(l1v)) * (l2n * np.sqrt(l2v)) * (gn * np.sqrt(gv))
    logp += np.sum(_normal_lp(w, 0.0, w_scale))
    logp += np.sum(_normal_lp(x_bias, x_bias_loc0, x_bias_scale0))
    if use_distortion:
        logp += np.sum(-np.log(np.pi * 0.1) - np.log1p(np.square(dist_c / 0.1)))
        x_loc = design @ (w + dist_c @ W) + x_bias
    else:
        x_loc = design @ w + x_bias
    logp += np.sum(np.log(2.0) - np.log(np.pi) - np.log1p(np.square(conc_c)))
    logp += np.sum(np.log(2.0) - np.log(np.pi) - np.log1p(np.square(scale_c)))
    conc = (conc_c[:, None] * W).sum(axis=0)
    scale = (scale_c[:, None] * W).sum(axis=0)
    logp += np.sum(_invgamma_lp(x_scale, conc, scale))
    logp += np.sum(_normal_lp(x, x_loc - np.asarray(sample_scales, np.float64).reshape(-1, 1), x_scale))
    if not use_point_estimates:
        m = p["qx_loc"].max(axis=1)
        t = m + np.log(np.exp(p["qx_loc"] - m[:, None]).sum(axis=1))
        logp += np.sum(_normal_lp(t, 0.0, scale_penalty))
        if lik is not None:
            logp += float(np.sum(lik(x)))
    z.update(x=x, w=w, x_bias=x_bias, x_scale=x_scale, x_loc=x_loc, w_scale=w_scale)
    return float(logq - logp), z


def data_statistics(p, eps, design, W, sample_scales, use_distortion, scale_penalty, lik=None):
    """What ONE rank's samples contribute to a step when the samples are sharded: the (F+2) x n
    sums of the observation model x ~ Normal(x_loc - scale_s, x_scale) that the shared parameters' gradients need, and
    the samples' own loss terms.  `p` / `eps` hold the shared entries and this rank's rows of qx_* / x.
    rows 0..F-1: sum_s F_sf * (-(x - mu)/x_scale^2); row F: sum_s (x - mu)/x_scale^2; row F+1: sum_s (1/x_scale -
    (x - mu)^2/x_scale^3).  Returns (stats [F+2, n], loss_of_samples)."""
    design = np.asarray(design, np.float64)
    w = p["qw_loc"] + softplus(p["qw_softplus_scale"]) * eps["w"]
    b = p["qx_bias_loc"] + softplus(p["qx_bias_softplus_scale"]) * eps["x_bias"]
    xs = softplus(p["qx_scale_loc"] + softplus(p["qx_scale_softplus_scale"]) * eps["x_scale"])
    sx = softplus(p["qx_softplus_scale"])
    x = p["qx_loc"] + sx * eps["x"]
    weff = w + (p["qw_distortion_c_loc"] @ W if use_distortion else 0.0)
    mu = design @ weff + b - np.asarray(sample_scales, np.float64).reshape(-1, 1)
    a = (x - mu) / xs ** 2
    stats = np.concatenate([-(design.T @ a), a.sum(axis=0, keepdims=True),
                            (1.0 / xs - (x - mu) ** 2 / xs ** 3).sum(axis=0, keepdims=True)])
    m = p["qx_loc"].max(axis=1)
    t = m + np.log(np.exp(p["qx_loc"] - m[:, None]).sum(axis=1))
    loss = -np.sum(_normal_lp(x, mu, xs)) + np.sum(_normal_lp(x, p["qx_loc"], sx)) - np.sum(_normal_lp(t, 0.0, scale_penalty))
    if lik is not None:
        loss -= float(np.sum(lik(x)))
    return stats, float(loss)


def adam_step(theta, g, m, v, t, lr=2e-3, b1=0.9, b2=0.999, eps=1e-7):
    """tf.optimizers.Adam (Keras): theta -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps), t = 1, 2, ..."""
    m[:] = b1 * m + (1 - b1) * g
    v[:] = b2 * v + (1 - b2) * g * g
    theta -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)


# ---- RNASeqJointLinearRegression (models/polee_regression.py:879-1283) ------------------------------------------------------
JOINT_TRANSCRIPT_PARAMS = [("qx_iso_scale_loc", "t"), ("qx_iso_scale_softplus_scale", "t"), ("qx_iso_loc", "St"),
                           ("qx_iso_softplus_scale", "St")]
JOINT_TRANSCRIPT_NOISE = [("x_iso_scale", "t"), ("x_iso", "St")]


def _halfcauchy_lp(x, scale):
    return np.log(2.0) - np.log(np.pi * scale) - np.log1p(np.square(x / scale))


def joint_regression_loss(p, eps, sp, se, tp, te, design, hinges, bandwidth, sample_scales, num_gene_features,
                          pair_transcript, pair_feature, gene_lik=None, frozen_weights=None):
    """loss = log q - log p of the joint model at one draw.
    p / eps: the GENE block (PARAMS / NOISE with n = gene features; the local2 and distortion entries are unused);
    sp / se: the SPLICE block (PARAMS / NOISE with n = P splice features, deg = 0, S = 0; local2 and x_scale unused);
    tp / te: the transcripts' part (JOINT_TRANSCRIPT_PARAMS / _NOISE).  pair_*: the feature matrix's non-zeros (0-based).
    model_fn :1007-1121, variational_model_fn :1124-1203.  Returns (loss, draws)."""
    design = np.asarray(design, np.float64)
    logq = 0.0

    def sp_normal(d, loc, sraw, e):
        nonlocal logq
        s = softplus(d[sraw])
        u = d[loc] + s * e
        logq += np.sum(_normal_lp(u, d[loc], s) - log_sigmoid(u))
        return softplus(u)

    def normal(d, loc, sraw, e):
        nonlocal logq
        s = softplus(d[sraw])
        v = d[loc] + s * e
        logq += np.sum(_normal_lp(v, d[loc], s))
        return v

    def horseshoe(d, e):
        gv = sp_normal(d, "qw_global_scale_variance_loc", "qw_global_scale_variance_softplus_scale", e["w_global_scale_variance"])
        gn = sp_normal(d, "qw_global_scale_noncentered_loc", "qw_global_scale_noncentered_softplus_scale", e["w_global_scale_noncentered"])
        lv = sp_normal(d, "qw_local1_scale_variance_loc", "qw_local1_scale_variance_softplus_scale", e["w_local1_scale_variance"])
        ln = sp_normal(d, "qw_local1_scale_noncentered_loc", "qw_local1_scale_noncentered_softplus_scale", e["w_local1_scale_noncentered"])
        w = normal(d, "qw_loc", "qw_softplus_scale", e["w"])
        lp = np.sum(_invgamma_lp(gv, 0.5, 0.5)) + np.sum(_halfnormal_lp(gn)) + np.sum(_invgamma_lp(lv, 0.5, 0.5)) + np.sum(_halfnormal_lp(ln))
        lp += np.sum(_normal_lp(w, 0.0, (ln * np.sqrt(lv)) * (gn * np.sqrt(gv))))
        return w, lp

    # ---- gene block
    w_gene, logp = horseshoe(p, eps)
    x_gene_bias = normal(p, "qx_bias_loc", "qx_bias_softplus_scale", eps["x_bias"])
    conc_c = softplus(p["qx_scale_concentration_c_loc"])
    scale_c = softplus(p["qx_scale_scale_c_loc"])
    x_scale = sp_normal(p, "qx_scale_loc", "qx_scale_softplus_scale", eps["x_scale"])
    x_gene = normal(p, "qx_loc", "qx_softplus_scale", eps["x"])
    logp += np.sum(_normal_lp(x_gene_bias, np.log(1.0 / num_gene_features), 12.0))
    W = kernel_regression_weights(bandwidth, x_gene_bias, np.asarray(hinges, np.float64))   # of the SAMPLED bias (:1034-1035)
    if frozen_weights is not None:     # (tests: what the gradient would be if the weights did not depend on the bias)
        W = frozen_weights
    logp += np.sum(_halfcauchy_lp(conc_c, 10.0)) + np.sum(_halfcauchy_lp(scale_c, 10.0))
    logp += np.sum(_invgamma_lp(x_scale, (conc_c[:, None] * W).sum(axis=0), (scale_c[:, None] * W).sum(axis=0)))
    x_gene_loc = design @ w_gene + x_gene_bias
    logp += np.sum(_normal_lp(x_gene, x_gene_loc - np.asarray(sample_scales, np.float64).reshape(-1, 1), x_scale))
    m = p["qx_loc"].max(axis=1)
    t = m + np.log(np.exp(p["qx_loc"] - m[:, None]).sum(axis=1))
    logp += np.sum(_normal_lp(t, 0.0, 5e-4))
    # ---- splice block
    w_splice, lp_s = horseshoe(sp, se)
    logp += lp_s
    x_splice_bias = normal(sp, "qx_bias_loc", "qx_bias_softplus_scale", se["x_bias"])
    logp += np.sum(_normal_lp(x_splice_bias, 0.0, 10.0))
    mu = design @ w_splice + x_splice_bias                       # [S, P]
    nt = tp["qx_iso_scale_loc"].shape[0]
    A = np.zeros((nt, mu.shape[1]))
    np.add.at(A, (np.asarray(pair_transcript), np.asarray(pair_feature)), 1.0)
    x_iso_loc = mu @ A.T                                          # [S, nt]
    x_iso_scale = sp_normal(tp, "qx_iso_scale_loc", "qx_iso_scale_softplus_scale", te["x_iso_scale"])
    x_iso = normal(tp, "qx_iso_loc", "qx_iso_softplus_scale", te["x_iso"])
    logp += np.sum(_halfcauchy_lp(x_iso_scale, 1.0))
    logp += np.sum(_normal_lp(x_iso, x_iso_loc, x_iso_scale[None, :]))
    if gene_lik is not None:
        logp += float(np.sum(gene_lik(x_gene, x_iso)))
    return float(logq - logp), dict(x_gene=x_gene, x_iso=x_iso, x_iso_loc=x_iso_loc, W=W, x_gene_bias=x_gene_bias)
