"""CPU: host-side pieces of the regression mirror (no device calls) and the float64 restatement's internal consistency."""
import math

import numpy as np
from scipy.stats import norm, t as tdist

from oracle import regression_ref as RR
from polee_amd import regression as R


def test_estimate_sample_scales_is_the_reference_formula():
    """src/PoleeModel.jl:82-89: median over the features whose median expression exceeds the upper quantile of
    (median profile - sample)."""
    rng = np.random.default_rng(0)
    base = rng.normal(-8, 2, size=500)
    shifts = np.array([0.0, 0.7, -0.4, 1.5])
    x = base[None, :] - shifts[:, None] + rng.normal(0, 0.01, size=(4, 500))
    ss = R.estimate_sample_scales(x)
    assert ss.shape == (4, 1)
    # the scale of a sample is its offset from the median profile, up to the noise
    np.testing.assert_allclose(ss[:, 0] - ss[0, 0], shifts - shifts[0], atol=0.02)
    xm = np.median(x, axis=0)
    hi = xm > np.quantile(xm, 0.95)
    np.testing.assert_allclose(ss[:, 0], np.median(xm[hi][None] - x[:, hi], axis=1), rtol=1e-6)


def test_minimum_effect_size_bisection():
    """src/regression.jl:604-622: bisection until P(|w| < delta) is within 0.1 % of the target; like the reference the
    value returned is the midpoint of the FINAL bracket (one halving past the accepted delta), so its own coverage
    is only close to the target."""
    for mu, sigma, target in [(0.0, 1.0, 0.9), (2.5, 0.3, 0.9), (-1.0, 2.0, 0.5)]:
        d = R.find_minimum_effect_size(mu, sigma, target)
        cov = norm.cdf(d, mu, sigma) - norm.cdf(-d, mu, sigma)
        assert abs(cov - target) / target <= 0.02
    assert abs(R.find_minimum_effect_size(0.0, 1.0, 0.9) - 1.6449) < 5e-2


def test_write_regression_effects_format(tmp_path):
    """src/regression.jl:625-685: log2 units, t_10 credible interval, probabilities of a minimum effect."""
    qw_loc = np.array([[0.5, -2.0], [0.0, 1.0]])
    qw_scale = np.array([[0.1, 0.5], [1.0, 0.2]])
    out = tmp_path / "effects.csv"
    R.write_regression_effects(str(out), ["a", "b"], "transcript_id", ["t1", "t2"], np.zeros(2), np.ones(2), qw_loc,
                               qw_scale, 0.05, 0.95, 1.5, 0.9, write_variational_posterior_params=True)
    lines = out.read_text().strip().split("\n")
    assert lines[0] == ("factor,transcript_id,min_effect_size,mean_effect_size,lower_credible,upper_credible,"
                        "prob_de,prob_down_de,prob_up_de,qx_bias_loc,qx_scale,qw_loc,qw_scale")
    assert len(lines) == 5
    row = lines[2].split(",")  # factor a, feature t2
    assert row[:2] == ["a", "t2"]
    ln2 = math.log(2)
    assert abs(float(row[3]) - (-2.0 / ln2)) < 1e-5
    assert abs(float(row[4]) - (tdist.ppf(0.05, 10) * 0.5 - 2.0) / ln2) < 1e-5
    down = tdist.cdf((-math.log(1.5) + 2.0) / 0.5, 10)
    assert abs(float(row[7]) - down) < 1e-5 and abs(float(row[6]) - down) < 1e-5


def test_parameter_tables_agree_and_restatement_is_smooth():
    assert R.PARAM_TABLE == RR.PARAMS
    rng = np.random.default_rng(1)
    S, F, n, deg = 3, 2, 11, 4
    x_init = rng.normal(-3, 1, size=(S, n))
    p = RR.initial_params(x_init, F, deg)
    assert p["qw_softplus_scale"].max() == 0.0 and p["qx_scale_loc"].min() == -0.5
    vec = RR.flatten(p, RR.PARAMS) + rng.normal(0, 0.2, size=4 + F * deg + 2 * deg + 10 * F * n + 4 * n + 2 * S * n)
    eps = RR.unflatten(rng.normal(size=2 + 5 * F * n + 2 * n + S * n), RR.NOISE, S, F, n, deg)
    mean = x_init.mean(axis=0)
    W = RR.kernel_regression_weights(1.0, mean, RR.choose_knots(mean.min(), mean.max(), deg))
    np.testing.assert_allclose(W.sum(axis=0), 1.0, rtol=1e-12)
    kw = dict(design=np.eye(S, F), W=W, sample_scales=np.zeros(S), x_bias_loc0=-2.0, x_bias_scale0=12.0,
              use_distortion=True, scale_penalty=1.0, use_point_estimates=False)
    l0, z = RR.regression_loss(RR.unflatten(vec, RR.PARAMS, S, F, n, deg), eps, **kw)
    assert np.isfinite(l0) and z["x"].shape == (S, n)
    # the scale-drift penalty is the only term that sees qx_loc other than through x: shifting qx_loc by c and x_bias
    # by the same c leaves the x ~ Normal(x_loc, x_scale) term unchanged
    pp = RR.unflatten(vec.copy(), RR.PARAMS, S, F, n, deg)
    pp["qx_loc"] = pp["qx_loc"] + 0.3
    pp["qx_bias_loc"] = pp["qx_bias_loc"] + 0.3
    l1, _ = RR.regression_loss(pp, eps, **kw)
    m = RR.unflatten(vec, RR.PARAMS, S, F, n, deg)["qx_loc"]
    t0 = np.log(np.exp(m).sum(axis=1))
    b0 = RR.unflatten(vec, RR.PARAMS, S, F, n, deg)["qx_bias_loc"] + \
        RR.softplus(RR.unflatten(vec, RR.PARAMS, S, F, n, deg)["qx_bias_softplus_scale"]) * eps["x_bias"]
    expected = 0.5 * (((t0 + 0.3) ** 2).sum() - (t0 ** 2).sum()) + \
        0.5 * ((((b0 + 0.3) + 2.0) / 12.0) ** 2 - ((b0 + 2.0) / 12.0) ** 2).sum()
    np.testing.assert_allclose(l1 - l0, expected, rtol=1e-8, atol=1e-8)


def test_density_restatements_agree_with_an_independent_library():
    """The regression oracle (oracle/regression_ref.py) restates TFP's log-densities -- library-defined, no reference test pins
    them ("parity unpinned").  SciPy's implementations of the same distributions are an independent statement of the
    same definitions: InverseGamma(concentration, scale), HalfNormal(1), HalfCauchy(0, scale), Normal; SoftplusNormal is the
    change of variables the restatement writes out (log q = Normal(u) - log sigmoid(u) at x = softplus(u)), checked against a
    numerical derivative of the CDF."""
    from scipy import stats
    from oracle import regression_ref as RR
    rng = np.random.default_rng(3)
    x = np.exp(rng.normal(0, 1.5, size=200))
    for a, b in ((0.5, 0.5), (0.001, 0.001), (3.2, 0.7), (12.0, 40.0)):
        np.testing.assert_allclose(RR._invgamma_lp(x, a, b), stats.invgamma.logpdf(x, a, scale=b), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(RR._halfnormal_lp(x), stats.halfnorm.logpdf(x), rtol=1e-12, atol=1e-12)
    for sc in (1.0, 10.0, 0.1):
        np.testing.assert_allclose(RR._halfcauchy_lp(x, sc), stats.halfcauchy.logpdf(x, scale=sc), rtol=1e-12, atol=1e-12)
    v = rng.normal(0, 3, size=200)
    np.testing.assert_allclose(RR._normal_lp(v, 0.7, 2.5), stats.norm.logpdf(v, 0.7, 2.5), rtol=1e-12, atol=1e-12)
    # SoftplusNormal(loc, scale): X = softplus(U), U ~ Normal(loc, scale): density of X by differentiating its CDF numerically
    loc, scale = 0.3, 0.8
    u = rng.normal(loc, scale, size=50)
    xs = RR.softplus(u)
    lq = RR._normal_lp(u, loc, scale) - RR.log_sigmoid(u)
    inv = lambda y: np.log(np.expm1(y))  # softplus^-1
    h = 1e-5
    num = (stats.norm.cdf(inv(xs + h), loc, scale) - stats.norm.cdf(inv(xs - h), loc, scale)) / (2 * h)
    np.testing.assert_allclose(lq, np.log(num), rtol=1e-5, atol=1e-6)


def test_relaxed_onehot_density_and_its_derivatives():
    """classify's surrogate for the latent design rows (models/polee_regression.py:372-376): RelaxedOneHotCategorical(T, logits).
    The restated density (Maddison et al. 2017, eq. 10) integrates to 1 over the simplex (K = 2: a one-dimensional quadrature),
    agrees with the density of y = softmax((logits + Gumbel) / T) found by a histogram of draws, and its two derivatives
    (at fixed y w.r.t. the logits; w.r.t. y) match central differences."""
    rng = np.random.default_rng(2)
    for T in (5.0, 0.7):
        logits = np.array([[0.3, -0.8]])
        t = np.linspace(1e-6, 1 - 1e-6, 400001)
        y = np.stack([t, 1 - t], axis=1)
        lq = R.relaxed_onehot_terms(np.repeat(logits, len(t), 0), y, T)[0]
        assert abs(np.trapz(np.exp(lq), t) - 1.0) < 2e-3
        g = -np.log(-np.log(rng.uniform(size=(400000, 2))))
        draws = R._softmax((logits + g) / T)[:, 0]
        hist, edges = np.histogram(draws, bins=40, range=(0, 1), density=True)
        mid = 0.5 * (edges[1:] + edges[:-1])
        want = np.exp(R.relaxed_onehot_terms(np.repeat(logits, 40, 0), np.stack([mid, 1 - mid], 1), T)[0])
        inner = np.zeros(40, bool)
        inner[3:37] = True
        inner &= want > 0.2  # (bins with at least ~2 000 of the 400 000 draws)
        assert inner.sum() >= 10
        np.testing.assert_allclose(hist[inner], want[inner], rtol=0.08)
    K, S = 4, 3
    logits = rng.normal(size=(S, K))
    y = R._softmax(rng.normal(size=(S, K)))
    T = 1.7
    lq, d_logits, d_y = R.relaxed_onehot_terms(logits, y, T)
    h = 1e-6
    for s_ in range(S):
        for k in range(K):
            e = np.zeros((S, K))
            e[s_, k] = h
            fd = (R.relaxed_onehot_terms(logits + e, y, T)[0][s_] - R.relaxed_onehot_terms(logits - e, y, T)[0][s_]) / (2 * h)
            assert abs(fd - d_logits[s_, k]) < 1e-6 * (1 + abs(fd))
            fd = (R.relaxed_onehot_terms(logits, y + e, T)[0][s_] - R.relaxed_onehot_terms(logits, y - e, T)[0][s_]) / (2 * h)
            assert abs(fd - d_y[s_, k]) < 1e-5 * (1 + abs(fd))
