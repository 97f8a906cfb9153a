"""-m gpu: the regression model's variational step (csrc/regression.hip) through the C ABI against the NumPy float64
restatement of models/polee_regression.py (oracle/regression_ref.py) and the oracle's approximate likelihood.

Tolerances: the loss is a sum of O(10 (F+S) n) float32 terms -> rtol 1e-4 (north_star's tolerance); gradients
against central finite differences of the float64 restatement (plus the oracle's analytic likelihood gradient) ->
rtol 2e-3 of the gradient's scale.
"""
import numpy as np
import pytest

from conftest import random_tree
from oracle import oracle as O
from oracle import regression_ref as RR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import polee_amd
    return polee_amd


@pytest.fixture(scope="module")
def ctx(P):
    return P.Context(0)


def _problem(rng, S, F, n):
    trees = [random_tree(n, rng) for _ in range(S)]
    idx = [O.make_inverse_ptt_params(*tr) for tr in trees]
    L_, R_, F_ = (np.stack([i[j] for i in idx]) for j in range(3))
    eff = rng.uniform(200, 3000, size=(S, n)).astype(np.float32)
    mu = rng.normal(0, 1, size=(S, n - 1)).astype(np.float32)
    sigma = np.exp(rng.normal(-1, 0.3, size=(S, n - 1))).astype(np.float32)
    alpha = rng.normal(0, 0.3, size=(S, n - 1)).astype(np.float32)
    vars_ = dict(efflen=eff, la_mu=mu, la_sigma=sigma, la_alpha=alpha, left_index=L_, right_index=R_, leaf_index=F_)
    design = np.zeros((S, F), np.float32)
    design[np.arange(S), np.arange(S) % F] = 1.0
    x_init = (rng.normal(-np.log(n), 1.5, size=(1, n)) + rng.normal(0, 0.4, size=(S, n))).astype(np.float32)
    return vars_, design, x_init


def _lik(vars_):
    a = (vars_["efflen"], vars_["la_mu"], vars_["la_sigma"], vars_["la_alpha"], vars_["left_index"],
         vars_["right_index"], vars_["leaf_index"])
    return (lambda x: O.approx_log_prob(x.astype(np.float32), *a).astype(np.float64),
            lambda x: O.approx_log_prob(x.astype(np.float32), *a, want_grad=True)[1].astype(np.float64))


def _oracle_setup(reg, design, x_init, ss, deg, bandwidth):
    mean = x_init.astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float64)
    hinges = RR.choose_knots(mean.min(), mean.max(), deg)
    return RR.kernel_regression_weights(bandwidth, mean, hinges)


@pytest.mark.parametrize("use_distortion,point,F,deg", [(True, False, 2, 5), (False, False, 2, 5), (True, True, 2, 5),
                                                         # the fixed-trip-count kernel instances (15 hinges, F <= 4)
                                                         (True, False, 1, 15), (True, False, 3, 15),
                                                         (True, False, 4, 15), (True, False, 5, 15)])
def test_loss_and_gradient_match_restatement(P, ctx, use_distortion, point, F, deg):
    rng = np.random.default_rng(31)
    S, n, pen = 4, 150, 0.7
    vars_, design, x_init = _problem(rng, S, F, n)
    ss = P.estimate_sample_scales(x_init, upper_quantile=0.8)
    reg = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, use_distortion, pen, point,
                                             kernel_regression_degree=deg, kernel_regression_bandwidth=1.3, ctx=ctx)
    W = _oracle_setup(reg, design, x_init, ss, deg, 1.3)
    np.testing.assert_allclose(reg.kernel_regression_weights(), W, rtol=2e-4, atol=1e-9)
    # initial values are the reference's
    p0 = RR.flatten(RR.initial_params(x_init, F, deg), RR.PARAMS)
    np.testing.assert_allclose(reg.get_flat_params(), p0, rtol=1e-6, atol=1e-6)
    # a generic point: perturb every parameter
    theta = (p0 + rng.normal(0, 0.3, size=p0.size)).astype(np.float32)
    reg.set_flat_params(theta)
    eps = rng.normal(size=reg.num_noise).astype(np.float32)
    loss, g = reg.loss_and_gradients(noise=eps)
    lik, lik_grad = _lik(vars_)
    common = dict(design=design.astype(np.float64), W=W, sample_scales=ss, x_bias_loc0=np.log(1.0 / n),
                  x_bias_scale0=12.0, use_distortion=use_distortion, scale_penalty=pen, use_point_estimates=point)
    e = RR.unflatten(eps.astype(np.float64), RR.NOISE, S, F, n, deg)

    def L_rest(vec, with_lik):
        pp = RR.unflatten(vec, RR.PARAMS, S, F, n, deg)
        return RR.regression_loss(pp, e, lik=lik if with_lik else None, **common)

    loss_o, z = L_rest(theta.astype(np.float64), True)
    assert abs(loss - loss_o) <= 1e-4 * abs(loss_o) + 1e-2, (loss, loss_o)
    # gradient: central differences of the float64 restatement without the likelihood + the oracle's analytic
    # likelihood gradient chained through x = qx_loc + softplus(qx_softplus_scale) eps
    t64 = theta.astype(np.float64)
    table = RR.unflatten(np.arange(t64.size), RR.PARAMS, S, F, n, deg)
    check_idx = set(range(4 + F * deg + 2 * deg))  # every shared parameter
    for name, _ in RR.PARAMS[7:]:
        ids = table[name].reshape(-1).astype(int)
        check_idx.update(rng.choice(ids, size=min(12, ids.size), replace=False).tolist())
    glik = None if point else lik_grad(z["x"])
    gscale = np.abs(g).max()
    worst = 0.0
    o_qx = int(table["qx_loc"].reshape(-1)[0])
    for i in sorted(check_idx):
        if point and i >= o_qx:  # qx_loc is not trainable with point estimates, qx_softplus_scale is unused
            continue
        h = 1e-4 * max(1.0, abs(t64[i]))
        tp, tm = t64.copy(), t64.copy()
        tp[i] += h
        tm[i] -= h
        fd = (L_rest(tp, False)[0] - L_rest(tm, False)[0]) / (2 * h)
        if not point:
            o_loc, o_s = int(table["qx_loc"].reshape(-1)[0]), int(table["qx_softplus_scale"].reshape(-1)[0])
            if o_loc <= i < o_loc + S * n:
                fd -= glik.reshape(-1)[i - o_loc]
            elif o_s <= i < o_s + S * n:
                k = i - o_s
                fd -= glik.reshape(-1)[k] * e["x"].reshape(-1)[k] / (1.0 + np.exp(-t64[i]))
        err = abs(g[i] - fd) / (abs(fd) + 2e-3 * gscale)
        worst = max(worst, err)
        assert err < 2e-3 * 5, (i, g[i], fd)
    if point:  # nothing about x is trained
        assert not np.any(g[o_qx:])
    if not use_distortion:
        assert not np.any(g[4:4 + F * deg])


def test_fit_trajectory_matches_restatement_adam(P, ctx):
    """A few Adam steps with supplied noise: device parameters follow the float64 restatement driven by the device's
    own gradients' oracle counterparts (finite differences are too slow here, so the restatement re-uses the device
    gradient of step t only through the check above; here we check the optimiser semantics)."""
    rng = np.random.default_rng(32)
    S, F, n, deg = 3, 2, 60, 4
    vars_, design, x_init = _problem(rng, S, F, n)
    ss = np.zeros((S, 1), np.float32)
    reg = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, True, 0.5, False, kernel_regression_degree=deg,
                                             ctx=ctx)
    steps = 4
    noise = rng.normal(size=(steps, reg.num_noise)).astype(np.float32)
    theta = reg.get_flat_params().astype(np.float64)
    m, v = np.zeros_like(theta), np.zeros_like(theta)
    shadow = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, True, 0.5, False,
                                                kernel_regression_degree=deg, ctx=ctx)
    losses = []
    for t in range(1, steps + 1):
        shadow.set_flat_params(theta.astype(np.float32))
        l, g = shadow.loss_and_gradients(noise=noise[t - 1])
        losses.append(l)
        RR.adam_step(theta, g.astype(np.float64), m, v, t)
    out = reg.fit(steps, noise=noise, return_trace=True)
    np.testing.assert_allclose(out[-1], losses, rtol=2e-5)
    np.testing.assert_allclose(reg.get_flat_params(), theta, rtol=2e-4, atol=2e-5)
    qx_loc, qw_loc, qw_scale, qx_bias, qx_scale = out[:5]
    assert qx_loc.shape == (S, n) and qw_loc.shape == (F, n) and qw_scale.shape == (F, n)
    assert qx_bias.shape == (n,) and qx_scale.shape == (n,)


def test_fit_recovers_planted_effect(P, ctx):
    """Property test at a moderate size with the device RNG: two groups of samples whose approximate likelihoods
    are centred on expression profiles that differ in a known set of transcripts -> the fitted effect (qw_loc of the
    second factor) is large there and shrunk elsewhere, and the loss decreases."""
    rng = np.random.default_rng(33)
    S, F, n = 8, 2, 400
    tree = random_tree(n, rng)
    li, ri, fi = O.make_inverse_ptt_params(*tree)
    to = O.PTT(*tree)
    base = rng.normal(0, 1.0, size=n)
    effect = np.zeros(n)
    planted = rng.choice(n, 30, replace=False)
    effect[planted] = rng.choice([-3.0, 3.0], size=30)
    design = np.zeros((S, F), np.float32)
    design[:, 0] = 1.0
    design[S // 2:, 1] = 1.0
    eff = np.full((S, n), 1000.0, np.float32)
    mus, x_init = [], []
    for s in range(S):
        logx = base + design[s, 1] * effect + rng.normal(0, 0.05, size=n)
        x = np.exp(logx - logx.max())
        x /= x.sum()
        y = to.inverse_transform(x.astype(np.float32))[0]
        y = np.clip(y, 1e-6, 1 - 1e-6)
        mus.append(np.log(y) - np.log1p(-y))
        x_init.append(np.log(x))
    mu = np.array(mus, np.float32)
    x_init = np.array(x_init, np.float32)
    vars_ = dict(efflen=eff, la_mu=mu, la_sigma=np.full((S, n - 1), 0.05, np.float32),
                 la_alpha=np.zeros((S, n - 1), np.float32), left_index=li[None], right_index=ri[None],
                 leaf_index=fi[None])
    ss = P.estimate_sample_scales(x_init)
    reg = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, True, 1.0, False, ctx=ctx)
    out = reg.fit(3000, seed=5, return_trace=True)
    qw_loc, trace = out[1], out[-1]
    assert np.all(np.isfinite(trace))
    assert trace[-200:].mean() < trace[:200].mean()
    w1 = qw_loc[1]
    others = np.setdiff1d(np.arange(n), planted)
    assert np.mean(np.sign(w1[planted]) == np.sign(effect[planted])) > 0.9
    assert np.median(np.abs(w1[planted])) > 5 * np.median(np.abs(w1[others]))


def test_design_gradient_matches_finite_differences_of_the_loss(P, ctx):
    """classify (models/polee_regression.py:342-413) needs d loss / d design: polee_regression_design_grad against central
    differences of the device's own loss over the design entries, same noise (float32 loss: a step of 1e-2 on entries of
    size 0.3..0.7, tolerance 2 % of the gradient's scale), with and without point estimates and distortion."""
    rng = np.random.default_rng(41)
    S, F, n = 4, 3, 300
    vars_, _, x_init = _problem(rng, S, F, n)
    ss = P.estimate_sample_scales(x_init)
    for point, dist in ((False, True), (True, False)):
        design = P.regression._softmax(rng.normal(size=(S, F))).astype(np.float32)
        reg = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, dist, 1.0, point, ctx=ctx, kernel_regression_degree=5)
        p0 = reg.get_flat_params()
        p0 += rng.normal(0, 0.05, size=p0.size).astype(np.float32)  # (away from the initial values)
        reg.set_flat_params(p0)
        noise = rng.normal(size=reg.num_noise).astype(np.float32)
        reg.set_design(design)
        reg.loss_and_gradients(noise)
        g = reg.design_gradient().astype(np.float64)
        fd = np.zeros_like(g)
        h = 1e-2
        for s_ in range(S):
            for f in range(F):
                d = design.astype(np.float64).copy()
                d[s_, f] += h
                reg.set_design(d)
                lp = reg.loss_and_gradients(noise)[0]
                d[s_, f] -= 2 * h
                reg.set_design(d)
                lm = reg.loss_and_gradients(noise)[0]
                fd[s_, f] = (lp - lm) / (2 * h)
        np.testing.assert_allclose(g, fd, rtol=0.02, atol=0.02 * np.abs(fd).max())


def test_classify_recovers_the_classes_of_held_out_samples(P, ctx):
    """classify (models/polee_regression.py:342-413) end to end: a model fitted on eight samples of two classes (one-hot design,
    a planted effect in 30 of 400 transcripts), then four held-out samples whose classes are unknown: the returned class
    probabilities put every sample in its class, with point estimates and with the approximate likelihood."""
    rng = np.random.default_rng(35)
    S, St, F, n = 8, 4, 2, 400
    tree = random_tree(n, rng)
    li, ri, fi = O.make_inverse_ptt_params(*tree)
    to = O.PTT(*tree)
    base = rng.normal(0, 1.0, size=n)
    effect = np.zeros(n)
    planted = rng.choice(n, 30, replace=False)
    effect[planted] = rng.choice([-3.0, 3.0], size=30)
    cls = np.array([0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 1, 0])
    mus, xs = [], []
    for s_ in range(S + St):
        logx = base + cls[s_] * effect + rng.normal(0, 0.05, size=n)
        x = np.exp(logx - logx.max())
        x /= x.sum()
        y = np.clip(to.inverse_transform(x.astype(np.float32))[0], 1e-6, 1 - 1e-6)
        mus.append(np.log(y) - np.log1p(-y))
        xs.append(np.log(x))
    mu, x_all = np.array(mus, np.float32), np.array(xs, np.float32)

    def vars_of(rows):
        k = len(rows)
        return dict(efflen=np.full((k, n), 1000.0, np.float32), la_mu=mu[rows], la_sigma=np.full((k, n - 1), 0.05, np.float32),
                    la_alpha=np.zeros((k, n - 1), np.float32), left_index=li[None], right_index=ri[None], leaf_index=fi[None])
    train, test = np.arange(S), np.arange(S, S + St)
    design = np.zeros((S, F), np.float32)
    design[np.arange(S), cls[:S]] = 1.0
    ss = P.estimate_sample_scales(x_all)
    reg = P.RNASeqTranscriptLinearRegression(vars_of(train), x_all[train], design, ss[train], True, 1.0, False, ctx=ctx)
    reg.fit(2500, seed=5)
    for point in (True, False):
        probs, trace = reg.classify(vars_of(test), x_all[test], ss[test], point, 1500, seed=9, return_trace=True)
        assert probs.shape == (St, F) and np.all(np.isfinite(trace))
        np.testing.assert_allclose(probs.sum(axis=1), 1.0, rtol=1e-6)
        assert np.array_equal(probs.argmax(axis=1), cls[test]), probs
        assert probs[np.arange(St), cls[test]].min() > 0.8, probs


def test_regression_argument_errors(P, ctx):
    rng = np.random.default_rng(34)
    vars_, design, x_init = _problem(rng, 2, 2, 20)
    with pytest.raises(P.PoleeError):  # too many hinges for the kernel
        P.RNASeqTranscriptLinearRegression(vars_, x_init, design, np.zeros(2), True, 1.0, False,
                                           kernel_regression_degree=100, ctx=ctx)
    with pytest.raises(ValueError):
        P.RNASeqTranscriptLinearRegression(vars_, x_init, design, np.zeros(3), True, 1.0, False, ctx=ctx)
    reg = P.RNASeqTranscriptLinearRegression(None, x_init, design, np.zeros(2), True, 1.0, True, ctx=ctx)
    with pytest.raises(ValueError):
        reg.loss_and_gradients(noise=np.zeros(3, np.float32))


def test_sample_sharding_statistics_reproduce_the_whole_model(P, ctx):
    """SURVEY 8(e) for the regression: samples sharded over ranks, one all-reduce of (F+2) n + 1 statistics.  Emulated
    on one GPU with the debug hooks: two handles holding half the samples each; their data-pass statistics summed on
    the host and fed to the prior pass give the loss and shared-parameter gradients of the whole model, and each
    shard's qx gradients are the whole model's rows."""
    rng = np.random.default_rng(35)
    S, F, n, deg = 4, 2, 200, 6
    vars_, design, x_init = _problem(rng, S, F, n)
    ss = P.estimate_sample_scales(x_init, upper_quantile=0.8)
    kw = dict(kernel_regression_degree=deg, ctx=ctx)
    whole = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, True, 0.8, False, **kw)
    theta = whole.get_flat_params()
    theta = (theta + rng.normal(0, 0.2, size=theta.size)).astype(np.float32)
    whole.set_flat_params(theta)
    eps = rng.normal(size=whole.num_noise).astype(np.float32)
    loss, g = whole.loss_and_gradients(noise=eps)
    shared_n = whole.num_params - 2 * S * n
    e_shared = whole.num_noise - S * n
    mean = x_init.astype(np.float64).mean(axis=0)
    tw = whole.unflatten(theta)
    stats, shards = 0.0, []
    for rows in (slice(0, 2), slice(2, 4)):
        v = {k: (a[rows] if a.shape[0] == S else a) for k, a in vars_.items()}
        sh = P.RNASeqTranscriptLinearRegression(v, x_init[rows], design[rows], ss[rows], True, 0.8, False,
                                                x_init_mean=mean, **kw)
        np.testing.assert_allclose(sh.kernel_regression_weights(), whole.kernel_regression_weights(), rtol=1e-6)
        th = np.concatenate([theta[:shared_n], tw["qx_loc"][rows].reshape(-1), tw["qx_softplus_scale"][rows].reshape(-1)])
        sh.set_flat_params(th)
        ez = np.concatenate([eps[:e_shared], eps[e_shared:].reshape(S, n)[rows].reshape(-1)])
        st = sh._data_pass(ez)
        stats = stats + st.astype(np.float64)
        shards.append((sh, rows))
    # the summed statistics are the float64 restatement's (rows: F sums for w, one for x_bias, one for x_scale)
    W = _oracle_setup(whole, design, x_init, ss, deg, 1.0)
    lik, _ = _lik(vars_)
    st_o, loss_o = RR.data_statistics(RR.unflatten(theta.astype(np.float64), RR.PARAMS, S, F, n, deg),
                                      RR.unflatten(eps.astype(np.float64), RR.NOISE, S, F, n, deg), design, W, ss, True,
                                      0.8, lik=lik)
    np.testing.assert_allclose(stats[:(F + 2) * n].reshape(F + 2, n), st_o, rtol=2e-4, atol=2e-4 * np.abs(st_o).max())
    assert abs(stats[(F + 2) * n:].sum() - loss_o) <= 1e-4 * abs(loss_o)
    gs = np.abs(g).max()
    for sh, rows in shards:
        l, gg = sh._prior_pass(stats.astype(np.float32))
        assert abs(l - loss) <= 2e-5 * abs(loss)
        np.testing.assert_allclose(gg[:shared_n], g[:shared_n], rtol=2e-4, atol=2e-5 * gs)
        k = 2 * n  # rows held by the shard
        whole_qx = whole.unflatten(g)
        np.testing.assert_allclose(gg[shared_n:shared_n + k].reshape(2, n), whole_qx["qx_loc"][rows], rtol=1e-5,
                                   atol=1e-6 * gs)
        np.testing.assert_allclose(gg[shared_n + k:].reshape(2, n), whole_qx["qx_softplus_scale"][rows], rtol=1e-5,
                                   atol=1e-6 * gs)


def test_regression_with_one_rank_communicator(P, ctx):
    """RCCL communicator of one rank: the all-reduce is skipped / identity and the fit is unchanged."""
    rng = np.random.default_rng(36)
    S, F, n = 3, 2, 90
    vars_, design, x_init = _problem(rng, S, F, n)
    ss = np.zeros(S, np.float32)
    comm = P.Comm(ctx, 1, 0)
    a = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, True, 1.0, False, ctx=ctx, comm=comm)
    b = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, True, 1.0, False, ctx=ctx)
    oa, ob = a.fit(25, seed=9, return_trace=True), b.fit(25, seed=9, return_trace=True)
    np.testing.assert_allclose(oa[-1], ob[-1], rtol=1e-5)
    np.testing.assert_allclose(oa[1], ob[1], rtol=1e-3, atol=1e-5)


def test_normal_likelihood_variant_matches_restatement(P, ctx):
    """RNASeqNormalTranscriptLinearRegression (models/polee_regression.py:490-531): loss and gradient against the
    float64 restatement with its Normal(log softmax(x), scale) likelihood, by central differences of the whole loss."""
    rng = np.random.default_rng(37)
    S, F, n, deg, pen = 3, 2, 130, 5, 0.9
    _, design, x_loc = _problem(rng, S, F, n)
    x_scale = np.exp(rng.normal(-1.0, 0.5, size=(S, n))).astype(np.float32)
    ss = P.estimate_sample_scales(x_loc, upper_quantile=0.8)
    reg = P.RNASeqNormalTranscriptLinearRegression(None, x_loc, x_scale, design, ss, True, pen,
                                                   kernel_regression_degree=deg, ctx=ctx)
    W = _oracle_setup(reg, design, x_loc, ss, deg, 1.0)
    p0 = reg.get_flat_params()
    theta = (p0 + rng.normal(0, 0.3, size=p0.size)).astype(np.float32)
    reg.set_flat_params(theta)
    eps = rng.normal(size=reg.num_noise).astype(np.float32)
    loss, g = reg.loss_and_gradients(noise=eps)
    e = RR.unflatten(eps.astype(np.float64), RR.NOISE, S, F, n, deg)
    v64, s64 = x_loc.astype(np.float64), x_scale.astype(np.float64)

    def lik(x):
        m = x.max(axis=1, keepdims=True)
        ls = x - (m + np.log(np.exp(x - m).sum(axis=1, keepdims=True)))
        return (-0.5 * np.square((v64 - ls) / s64) - np.log(s64) - 0.5 * RR.LOG2PI).sum(axis=1)

    def Lf(vec):
        return RR.regression_loss(RR.unflatten(vec, RR.PARAMS, S, F, n, deg), e, design=design.astype(np.float64), W=W,
                                  sample_scales=ss, x_bias_loc0=np.log(1.0 / n), x_bias_scale0=12.0, use_distortion=True,
                                  scale_penalty=pen, use_point_estimates=False, lik=lik)[0]

    t64 = theta.astype(np.float64)
    lo = Lf(t64)
    assert abs(loss - lo) <= 1e-4 * abs(lo) + 1e-2
    table = RR.unflatten(np.arange(t64.size), RR.PARAMS, S, F, n, deg)
    gscale = np.abs(g).max()
    for name in ("qx_loc", "qx_softplus_scale", "qw_loc", "qx_bias_loc"):
        for i in rng.choice(table[name].reshape(-1).astype(int), size=10, replace=False):
            h = 1e-4 * max(1.0, abs(t64[i]))
            tp, tm = t64.copy(), t64.copy()
            tp[i] += h
            tm[i] -= h
            fd = (Lf(tp) - Lf(tm)) / (2 * h)
            assert abs(g[i] - fd) / (abs(fd) + 2e-3 * gscale) < 1e-2, (name, i, g[i], fd)
    out = reg.fit(50, seed=3, return_trace=True)
    assert np.all(np.isfinite(out[-1]))
    with pytest.raises(P.PoleeError):
        P.RNASeqNormalTranscriptLinearRegression(None, x_loc, -x_scale, design, ss, True, pen, ctx=ctx)


def test_gene_level_model_matches_restatement(P, ctx):
    """RNASeqGeneLinearRegression (models/polee_regression.py:533-600): loss against the float64 restatement with the
    C oracle's gene-level likelihood; gradients of the isoform block and of qx_loc against central differences of the
    restatement + the oracle's analytic likelihood gradients."""
    rng = np.random.default_rng(38)
    S, F, nt, G, deg, pen = 3, 2, 160, 45, 5, 0.8
    vars_, design, _ = _problem(rng, S, F, nt)
    gene_of = np.concatenate([np.arange(G), rng.integers(0, G, nt - G)])
    rng.shuffle(gene_of)
    x_gene_init = (rng.normal(-np.log(G), 1.2, size=(1, G)) + rng.normal(0, 0.3, size=(S, G))).astype(np.float32)
    x_iso_init = rng.normal(0, 1.0, size=(S, nt)).astype(np.float32)
    ss = P.estimate_sample_scales(x_gene_init, upper_quantile=0.7)
    reg = P.RNASeqGeneLinearRegression(vars_, gene_of + 1, np.arange(1, nt + 1), x_gene_init, x_iso_init, None, design, ss,
                                       True, pen, False, kernel_regression_degree=deg, ctx=ctx)
    n_iso_noise = nt + S * nt
    assert reg.num_isoform_params == 2 * nt + 2 * S * nt and reg.num_noise == 2 + 5 * F * G + 2 * G + S * G + n_iso_noise
    iv = reg.isoform_variables()
    np.testing.assert_allclose(iv["qx_isoform_mean_loc"], x_iso_init.mean(axis=0), rtol=1e-5, atol=1e-6)
    assert np.all(iv["qx_isoform_softplus_scale"] == -2.0) and np.all(iv["qx_isoform_mean_softplus_scale"] == -2.0)
    theta = (reg.get_flat_params() + rng.normal(0, 0.2, size=reg.num_params)).astype(np.float32)
    itheta = (reg.get_isoform_params() + rng.normal(0, 0.2, size=reg.num_isoform_params)).astype(np.float32)
    reg.set_flat_params(theta)
    reg.set_isoform_params(itheta)
    eps = rng.normal(size=reg.num_noise).astype(np.float32)
    loss, g = reg.loss_and_gradients(noise=eps)
    gi = reg.isoform_gradients()
    W = _oracle_setup(reg, design, x_gene_init, ss, deg, 1.0)
    e = RR.unflatten(eps[:-n_iso_noise].astype(np.float64), RR.NOISE, S, F, G, deg)
    ie = RR.unflatten_iso(eps[-n_iso_noise:].astype(np.float64), RR.ISO_NOISE, S, nt)
    a = (vars_["efflen"], vars_["la_mu"], vars_["la_sigma"], vars_["la_alpha"], vars_["left_index"],
         vars_["right_index"], vars_["leaf_index"])
    common = dict(design=design.astype(np.float64), W=W, sample_scales=ss, x_bias_loc0=np.log(1.0 / G),
                  x_bias_scale0=12.0, use_distortion=True, scale_penalty=pen, use_point_estimates=False)

    def base(vec):
        return RR.regression_loss(RR.unflatten(vec, RR.PARAMS, S, F, G, deg), e, lik=None, **common)

    def iso(ivec, x_gene, with_lik):
        lik = (lambda xg, xi: O.approx_gene_log_prob(xg.astype(np.float32), xi.astype(np.float32), gene_of, *a)) \
            if with_lik else None
        return RR.isoform_terms(RR.unflatten_iso(ivec, RR.ISO_PARAMS, S, nt), ie, x_gene, lik)

    t64, i64 = theta.astype(np.float64), itheta.astype(np.float64)
    lb, z = base(t64)
    li, xi = iso(i64, z["x"], True)
    assert abs(loss - (lb + li)) <= 1e-4 * abs(lb + li) + 1e-2, (loss, lb + li)
    _, gg, gxi = O.approx_gene_log_prob(z["x"].astype(np.float32), xi.astype(np.float32), gene_of, *a, want_grad=True)
    # isoform block: central differences of the restatement without the likelihood + analytic chain of the likelihood
    o_loc, o_s = 2 * nt, 2 * nt + S * nt
    scale = np.abs(gi).max()
    for i in np.concatenate([rng.choice(2 * nt, 12, replace=False), o_loc + rng.choice(S * nt, 12, replace=False),
                             o_s + rng.choice(S * nt, 12, replace=False)]):
        h = 1e-4 * max(1.0, abs(i64[i]))
        tp, tm = i64.copy(), i64.copy()
        tp[i] += h
        tm[i] -= h
        fd = (iso(tp, z["x"], False)[0] - iso(tm, z["x"], False)[0]) / (2 * h)
        if o_loc <= i < o_s:
            fd -= gxi.reshape(-1)[i - o_loc]
        elif i >= o_s:
            k = i - o_s
            fd -= gxi.reshape(-1)[k] * ie["x_isoform"].reshape(-1)[k] / (1.0 + np.exp(-i64[i]))
        assert abs(gi[i] - fd) / (abs(fd) + 2e-3 * scale) < 1e-2, (i, gi[i], fd)
    # gene-level qx_loc sees the likelihood through d lp / d x_gene
    table = RR.unflatten(np.arange(t64.size), RR.PARAMS, S, F, G, deg)
    o_qx = int(table["qx_loc"].reshape(-1)[0])
    gs = np.abs(g).max()
    for k in rng.choice(S * G, 12, replace=False):
        i = o_qx + k
        h = 1e-4 * max(1.0, abs(t64[i]))
        tp, tm = t64.copy(), t64.copy()
        tp[i] += h
        tm[i] -= h
        fd = (base(tp)[0] - base(tm)[0]) / (2 * h) - gg.reshape(-1)[k]
        assert abs(g[i] - fd) / (abs(fd) + 2e-3 * gs) < 1e-2, (i, g[i], fd)
    out = reg.fit(60, seed=4, return_trace=True)
    assert np.all(np.isfinite(out[-1])) and out[-1][-10:].mean() < out[-1][:10].mean()
    assert not np.allclose(reg.get_isoform_params(), itheta)


def test_gene_isoform_model_matches_restatement(P, ctx):
    """RNASeqGeneIsoformLinearRegression (models/polee_regression.py:656-877): loss against the float64 restatement
    (gene block + isoform regression block) with the C oracle's gene-level likelihood; EVERY gradient of the isoform
    block's shared parameters and samples of its per-sample ones against central differences of the restatement + the
    oracle's analytic likelihood gradients; initial values and fit()'s outputs as the reference's."""
    rng = np.random.default_rng(41)
    S, F, Fi, nt, G, deg, pen = 4, 2, 3, 120, 30, 5, 0.8
    vars_, design, _ = _problem(rng, S, F, nt)
    design_iso = np.concatenate([np.ones((S, 1)), rng.normal(0, 1, size=(S, Fi - 1))], axis=1).astype(np.float32)
    gene_of = np.concatenate([np.arange(G), rng.integers(0, G, nt - G)])
    rng.shuffle(gene_of)
    x_gene_init = (rng.normal(-np.log(G), 1.2, size=(1, G)) + rng.normal(0, 0.3, size=(S, G))).astype(np.float32)
    x_iso_init = rng.normal(0, 1.0, size=(S, nt)).astype(np.float32)
    ss = P.estimate_sample_scales(x_gene_init, upper_quantile=0.7)
    reg = P.RNASeqGeneIsoformLinearRegression(vars_, gene_of + 1, np.arange(1, nt + 1), x_gene_init, x_iso_init, None,
                                              design, design_iso, ss, True, pen, False, kernel_regression_degree=deg,
                                              ctx=ctx)
    n_ip = 4 + 10 * Fi * nt + 4 * nt + 2 * S * nt
    n_ie = 2 + 5 * Fi * nt + 2 * nt + S * nt
    assert reg.num_isoform_params == n_ip and reg.num_noise == 2 + 5 * F * G + 2 * G + S * G + n_ie
    iv = reg.isoform_variables()
    # initial values (models/polee_regression.py:736-775)
    np.testing.assert_allclose(iv["qx_isoform_bias_loc"], x_iso_init.mean(axis=0), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(iv["qx_isoform_loc"], x_iso_init)
    assert np.all(iv["qx_isoform_softplus_scale"] == -2.0) and np.all(iv["qx_isoform_scale_loc"] == 1.0)
    assert np.all(iv["qw_isoform_loc"] == 0.0) and np.all(iv["qw_isoform_softplus_scale"] == -1.0)
    for k, val in iv.items():
        if k.endswith("softplus_scale") and k != "qx_isoform_softplus_scale":
            assert np.all(val == -1.0), k
        elif "scale_variance_loc" in k or "scale_noncentered_loc" in k:
            assert np.all(val == 0.0), k
    theta = (reg.get_flat_params() + rng.normal(0, 0.2, size=reg.num_params)).astype(np.float32)
    itheta = (reg.get_isoform_params() + rng.normal(0, 0.2, size=n_ip)).astype(np.float32)
    reg.set_flat_params(theta)
    reg.set_isoform_params(itheta)
    eps = rng.normal(size=reg.num_noise).astype(np.float32)
    loss, g = reg.loss_and_gradients(noise=eps)
    gi = reg.isoform_gradients()
    assert np.all(np.isfinite(gi)) and np.all(np.isfinite(g))
    W = _oracle_setup(reg, design, x_gene_init, ss, deg, 1.0)
    e = RR.unflatten(eps[:-n_ie].astype(np.float64), RR.NOISE, S, F, G, deg)
    ie = RR.unflatten(eps[-n_ie:].astype(np.float64), RR.NOISE, S, Fi, nt, 0)
    a = (vars_["efflen"], vars_["la_mu"], vars_["la_sigma"], vars_["la_alpha"], vars_["left_index"],
         vars_["right_index"], vars_["leaf_index"])
    common = dict(design=design.astype(np.float64), W=W, sample_scales=ss, x_bias_loc0=np.log(1.0 / G),
                  x_bias_scale0=12.0, use_distortion=True, scale_penalty=pen, use_point_estimates=False)

    def base(vec):
        return RR.regression_loss(RR.unflatten(vec, RR.PARAMS, S, F, G, deg), e, lik=None, **common)

    def iso(ivec, x_gene, with_lik):
        lik = (lambda xg, xi: O.approx_gene_log_prob(xg.astype(np.float32), xi.astype(np.float32), gene_of, *a)) \
            if with_lik else None
        return RR.isoform_regression_terms(RR.unflatten(ivec, RR.PARAMS, S, Fi, nt, 0), ie, design_iso, x_gene, lik)

    t64, i64 = theta.astype(np.float64), itheta.astype(np.float64)
    lb, z = base(t64)
    li, xi = iso(i64, z["x"], True)
    assert abs(loss - (lb + li)) <= 1e-4 * abs(lb + li) + 1e-2, (loss, lb + li)
    _, gg, gxi = O.approx_gene_log_prob(z["x"].astype(np.float32), xi.astype(np.float32), gene_of, *a, want_grad=True)
    # The likelihood reaches the isoform block through x_isoform alone: d loss / d theta = d(restatement without the
    # likelihood) / d theta - <d lp / d x_isoform, d x_isoform / d theta>, the second factor by differences too.
    o_loc, o_s = n_ip - 2 * S * nt, n_ip - S * nt
    scale = np.abs(gi[:o_loc]).max()
    shared = np.concatenate([np.arange(4), 4 + rng.choice(10 * Fi * nt + 4 * nt, 60, replace=False)])
    own = np.concatenate([o_loc + rng.choice(S * nt, 15, replace=False), o_s + rng.choice(S * nt, 15, replace=False)])
    worst = 0.0
    for i in np.concatenate([shared, own]):
        h = 1e-4 * max(1.0, abs(i64[i]))
        tp, tm = i64.copy(), i64.copy()
        tp[i] += h
        tm[i] -= h
        (lp_, xp), (lm_, xm) = iso(tp, z["x"], False), iso(tm, z["x"], False)
        fd = (lp_ - lm_) / (2 * h) - float(np.sum(gxi * (xp - xm))) / (2 * h)
        err = abs(gi[i] - fd) / (abs(fd) + 2e-3 * (scale if i < o_loc else np.abs(gi[o_loc:]).max()))
        worst = max(worst, err)
        assert err < 1e-2, (i, gi[i], fd)
    # gene-level qx_loc sees the likelihood through d lp / d x_gene
    table = RR.unflatten(np.arange(t64.size), RR.PARAMS, S, F, G, deg)
    o_qx = int(table["qx_loc"].reshape(-1)[0])
    gs = np.abs(g).max()
    for k in rng.choice(S * G, 12, replace=False):
        i = o_qx + k
        h = 1e-4 * max(1.0, abs(t64[i]))
        tp, tm = t64.copy(), t64.copy()
        tp[i] += h
        tm[i] -= h
        fd = (base(tp)[0] - base(tm)[0]) / (2 * h) - gg.reshape(-1)[k]
        assert abs(g[i] - fd) / (abs(fd) + 2e-3 * gs) < 1e-2, (i, g[i], fd)
    out = reg.fit(60, seed=4, return_trace=True)
    assert len(out) == 11 and np.all(np.isfinite(out[-1])) and out[-1][-10:].mean() < out[-1][:10].mean()
    assert out[2].shape == (Fi, nt) and out[3].shape == (Fi, nt) and out[4].shape == (nt,) and out[6].shape == (nt,)
    assert out[9].shape == (S, G)
    assert not np.allclose(reg.get_isoform_params(), itheta)


def test_joint_model_matches_restatement(P, ctx):
    """RNASeqJointLinearRegression (models/polee_regression.py:879-1283): the loss against the float64 restatement (gene
    block with a horseshoe prior, kernel-regression weights of the SAMPLED bias and HalfCauchy(0, 10) coefficients; splice
    block over P features aggregated to the transcripts; HalfCauchy x_iso_scale) with the C oracle's gene-level likelihood,
    and gradients of every kind of parameter against central differences of the restatement (+ the oracle's analytic
    likelihood gradients); initial values; fit()'s outputs."""
    rng = np.random.default_rng(53)
    S, F, nt, G, Pf, deg = 4, 2, 90, 25, 30, 5
    vars_, design, _ = _problem(rng, S, F, nt)
    gene_of = np.concatenate([np.arange(G), rng.integers(0, G, nt - G)])
    rng.shuffle(gene_of)
    # splice features: every transcript takes part in 0..3 of them, every feature has at least one transcript
    pt, pf = [], []
    for t in range(nt):
        for f in rng.choice(Pf, rng.integers(0, 4), replace=False):
            pt.append(t); pf.append(int(f))
    for f in range(Pf):
        if f not in pf:
            pt.append(int(rng.integers(0, nt))); pf.append(f)
    pt, pf = np.array(pt), np.array(pf)
    x_gene_init = (rng.normal(-np.log(G), 1.2, size=(1, G)) + rng.normal(0, 0.3, size=(S, G))).astype(np.float32)
    x_iso_init = rng.normal(0, 1.0, size=(S, nt)).astype(np.float32)
    ss = P.estimate_sample_scales(x_gene_init, upper_quantile=0.7)
    reg = P.RNASeqJointLinearRegression(vars_, np.arange(1, nt + 1), gene_of + 1, G, pt + 1, pf + 1, Pf, None, x_gene_init,
                                        x_iso_init, design, ss, False, kernel_regression_degree=deg, ctx=ctx)
    n_sp = 4 + 10 * F * Pf + 4 * Pf
    n_ip = n_sp + 2 * nt + 2 * S * nt
    n_se = 2 + 5 * F * Pf + 2 * Pf
    n_ie = n_se + nt + S * nt
    assert reg.num_isoform_params == n_ip and reg.num_noise == 2 + 5 * F * G + 2 * G + S * G + n_ie
    sv, gv = reg.splice_variables(), reg.variables()
    # initial values (:925-1010)
    assert np.all(sv["qx_iso_scale_loc"] == 3.0) and np.all(sv["qx_iso_scale_softplus_scale"] == -1.0)
    np.testing.assert_array_equal(sv["qx_iso_loc"], x_iso_init)
    assert np.all(sv["qx_iso_softplus_scale"] == -3.0) and np.all(sv["qw_splice_softplus_scale"] == -2.0)
    assert np.all(sv["qw_splice_loc"] == 0.0) and np.all(sv["qx_splice_bias_loc"] == 0.0) and np.all(sv["qx_splice_bias_softplus_scale"] == -1.0)
    assert np.all(gv["qw_softplus_scale"] == -2.0) and np.all(gv["qx_scale_loc"] == -0.5)
    np.testing.assert_allclose(gv["qx_bias_loc"], x_gene_init.mean(axis=0), rtol=1e-5, atol=1e-6)
    theta = (reg.get_flat_params() + rng.normal(0, 0.2, size=reg.num_params)).astype(np.float32)
    itheta = (reg.get_isoform_params() + rng.normal(0, 0.2, size=n_ip)).astype(np.float32)
    reg.set_flat_params(theta)
    reg.set_isoform_params(itheta)
    eps = rng.normal(size=reg.num_noise).astype(np.float32)
    loss, g = reg.loss_and_gradients(noise=eps)
    gi = reg.isoform_gradients()
    assert np.all(np.isfinite(gi)) and np.all(np.isfinite(g))
    mean0 = x_gene_init.astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float64)
    hinges = RR.choose_knots(mean0.min(), mean0.max(), deg)
    e = RR.unflatten(eps[:-n_ie].astype(np.float64), RR.NOISE, S, F, G, deg)
    se = RR.unflatten(eps[-n_ie:-n_ie + n_se].astype(np.float64), RR.NOISE, 0, F, Pf, 0)
    te = RR.unflatten_iso(eps[-n_ie + n_se:].astype(np.float64), RR.JOINT_TRANSCRIPT_NOISE, S, nt)
    a = (vars_["efflen"], vars_["la_mu"], vars_["la_sigma"], vars_["la_alpha"], vars_["left_index"],
         vars_["right_index"], vars_["leaf_index"])

    def model(tvec, ivec, with_lik, frozen=None):
        lik = (lambda xg, xi: O.approx_gene_log_prob(xg.astype(np.float32), xi.astype(np.float32), gene_of, *a)) if with_lik else None
        return RR.joint_regression_loss(RR.unflatten(tvec, RR.PARAMS, S, F, G, deg), e,
                                        RR.unflatten(ivec[:n_sp], RR.PARAMS, 0, F, Pf, 0), se,
                                        RR.unflatten_iso(ivec[n_sp:], RR.JOINT_TRANSCRIPT_PARAMS, S, nt), te,
                                        design, hinges, 1.0, ss, G, pt, pf, lik, frozen)

    t64, i64 = theta.astype(np.float64), itheta.astype(np.float64)
    lref, z = model(t64, i64, True)
    assert abs(loss - lref) <= 1e-4 * abs(lref) + 1e-2, (loss, lref)
    _, gg, gxi = O.approx_gene_log_prob(z["x_gene"].astype(np.float32), z["x_iso"].astype(np.float32), gene_of, *a, want_grad=True)

    def fd(tvec, ivec, which, i, frozen=None):
        h = 1e-4 * max(1.0, abs((tvec if which == 0 else ivec)[i]))
        vp, vm = (tvec.copy(), ivec.copy()), (tvec.copy(), ivec.copy())
        vp[which][i] += h
        vm[which][i] -= h
        (lp_, zp), (lm_, zm) = model(vp[0], vp[1], False, frozen), model(vm[0], vm[1], False, frozen)
        # the likelihood reaches the parameters through x_gene and x_iso alone
        return ((lp_ - lm_) - float(np.sum(gg * (zp["x_gene"] - zm["x_gene"]))) - float(np.sum(gxi * (zp["x_iso"] - zm["x_iso"])))) / (2 * h)

    # gene block: globals, mean-variance coefficients, per-column arrays (local1, w, bias -- the bias also through the
    # weights --, x_scale), per-sample qx; the local2 and distortion entries are unused: gradient 0
    table = RR.unflatten(np.arange(t64.size), RR.PARAMS, S, F, G, deg)
    pick = lambda name, k: rng.choice(table[name].reshape(-1).astype(int), min(k, table[name].size), replace=False)
    # (tolerances scale with the gradients of the SAME array: the 5e-4 drift penalty makes qx_loc's far larger than the rest)
    groups = [("globals", np.arange(4)), ("qx_scale_concentration_c_loc", table["qx_scale_concentration_c_loc"].astype(int)),
              ("qx_scale_scale_c_loc", table["qx_scale_scale_c_loc"].astype(int))]
    groups += [(nm, pick(nm, 6)) for nm in ("qw_local1_scale_variance_loc", "qw_local1_scale_noncentered_softplus_scale", "qw_loc",
                                            "qw_softplus_scale", "qx_bias_loc", "qx_bias_softplus_scale", "qx_scale_loc",
                                            "qx_scale_softplus_scale", "qx_loc", "qx_softplus_scale")]
    for nm, idx in groups:
        gs = np.abs(g[idx]).max()
        for i in idx:
            f_ = fd(t64, i64, 0, int(i))
            assert abs(g[i] - f_) / (abs(f_) + 2e-3 * gs) < 1e-2, ("gene block", nm, int(i), g[i], f_)
    for nm in ("qw_local2_scale_variance_loc", "qw_local2_scale_noncentered_loc", "qw_distortion_c_loc"):
        assert np.all(g[table[nm].reshape(-1).astype(int)] == 0.0), nm
    # the weights are a function of the SAMPLED bias (:1034-1035): the check above has the power to see that path --
    # with the weights frozen the restatement's bias gradients differ from the device's by far more than the tolerance
    idx = table["qx_bias_loc"].astype(int)
    gs = np.abs(g[idx]).max()
    frozen_err = [abs(g[i] - fd(t64, i64, 0, int(i), z["W"])) / (abs(g[i]) + 2e-3 * gs) for i in idx]
    assert np.sum(np.array(frozen_err) > 5e-2) >= 3, sorted(frozen_err)[-5:]
    # splice block + transcripts
    stab = RR.unflatten(np.arange(n_sp), RR.PARAMS, 0, F, Pf, 0)
    spick = lambda name, k: rng.choice(stab[name].reshape(-1).astype(int), k, replace=False)
    o_t = n_sp
    groups = [("globals", np.arange(4))]
    groups += [(nm, spick(nm, 6)) for nm in ("qw_local1_scale_variance_loc", "qw_local1_scale_noncentered_loc", "qw_loc",
                                             "qw_softplus_scale", "qx_bias_loc", "qx_bias_softplus_scale")]
    groups += [("x_iso_scale loc", o_t + rng.choice(nt, 6, replace=False)), ("x_iso_scale s", o_t + nt + rng.choice(nt, 6, replace=False)),
               ("x_iso loc", o_t + 2 * nt + rng.choice(S * nt, 8, replace=False)),
               ("x_iso s", o_t + 2 * nt + S * nt + rng.choice(S * nt, 8, replace=False))]
    for nm, idx in groups:
        gis = np.abs(gi[idx]).max()
        for i in idx:
            f_ = fd(t64, i64, 1, int(i))
            assert abs(gi[i] - f_) / (abs(f_) + 2e-3 * gis) < 1e-2, ("splice block", nm, int(i), gi[i], f_)
    for nm in ("qw_local2_scale_variance_loc", "qx_scale_loc", "qx_scale_softplus_scale"):
        assert np.all(gi[stab[nm].reshape(-1).astype(int)] == 0.0), nm
    out = reg.fit(80, seed=4, return_trace=True)
    assert len(out) == 5 and np.all(np.isfinite(out[-1])) and out[-1][-10:].mean() < out[-1][:10].mean()
    assert out[0].shape == (F, G) and out[1].shape == (F, G) and out[2].shape == (F, Pf) and out[3].shape == (F, Pf)
    assert not np.allclose(reg.get_isoform_params(), itheta)
