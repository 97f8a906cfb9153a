"""Tighter pins of the CPU oracle on the reference's own fixtures.

(a) Stationarity: prep.h5 holds the parameters the REFERENCE fitted to likelihood-matrix.h5.  If the oracle's ELBO
    gradient is the reference's, those parameters are (nearly) stationary for it: the mean gradient over many draws is
    what ADAM's jitter leaves.  A missing or biased term -- one of the ladj gradients of logitnormal.jl:50-53 /
    sinh_arcsinh.jl:35-36 / ptt.jl:197-204, the effective-length Jacobian of likelihood.jl:102-104 -- adds its own
    expected pattern over the nodes to that mean gradient: the test regresses the mean gradient on each term's pattern
    (weights = inverse gradient variance) and has a measured power: a bias of 3 % (4 %, 7 % for two of the six terms),
    injected into the oracle through a debug scale, makes it fail.
(b) Change of variables: for x = log(tf sampler(z0)) the density of the fitted approximation (the TF side,
    polee_approx_likelihood.py:367-450) equals the standard-normal density of z0 minus the log-determinants of the
    Julia-side forward transforms (sinh_arcsinh.jl:10-23, logitnormal.jl:8-20, ptt.jl:125-160) plus the two terms the
    TF code adds for the softmax and the effective lengths.  This ties rows a25 / a8 / a9 to the forward path that the
    reference fixtures pin.
"""
import numpy as np

from oracle import oracle as O


def _fixture(lm_fixture, prep_fixture):
    f = lm_fixture
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    return f, so, to


# The six additive terms of the ELBO gradient that the stationarity pin can tell apart (oracle_debug_scale in
# polee_oracle.c): their reference lines, and the smallest relative bias of each that makes the test below FAIL
# whatever its sign (the documented detection threshold; round 3's test accepted a mean gradient up to 15 % of a term).
TERMS = [
    ("logitnormal.jl:50     mu    += 1 - 2y", 0.03),
    ("logitnormal.jl:51-52  sigma += 1/sigma + zs (1 - 2y)", 0.04),
    ("logitnormal.jl:53     z     += sigma (1 - 2y)", 0.03),
    ("sinh_arcsinh.jl:36    alpha += tanh(c)", 0.07),
    ("ptt.jl:203-204        ladj part of transform_gradients!", 0.03),
    ("likelihood.jl:102-104 effective-length Jacobian", 0.035),
]
# |beta_j| must stay below this for the test to pass (what the reference's own ADAM residual leaves, measured: -0.006,
# -0.015, -0.006, +0.005, +0.005, +0.006, with standard errors 0.004 .. 0.013)
BETA_MAX = np.array([0.02, 0.024, 0.02, 0.045, 0.02, 0.024])


def term_patterns(so, to, f, p, ndraws, seed):
    """E[contribution of term j to (mu, omega, alpha) gradients] [6, 3 (n-1)]: the oracle's mean gradient with the term's
    debug scale at 2 minus the same draws' mean gradient with it at 1 (every term enters the gradients linearly)."""
    args = (so, to, f["effective_lengths"], p["mu"], p["omega"], p["alpha"], seed, ndraws)
    base, _ = O.vi_pin_stats(*args)
    T = []
    for j in range(6):
        O.set_debug_scale(j, 2.0)
        try:
            b, _ = O.vi_pin_stats(*args)
        finally:
            O.set_debug_scale(j, 1.0)
        T.append((b - base).ravel())
    return np.array(T)


def term_betas(gbar, sd, T):
    """Per term: the weighted least-squares coefficient of the mean gradient on the term's pattern, weights 1 / sd^2
    (sd = per-draw standard deviation of a node's gradient: ADAM's stationary jitter scales with it).  If the oracle's
    term j were (1 + b) times the reference's, the reference's fitted parameters would leave E[g] = residual + b T_j:
    beta_j moves by exactly b."""
    y, w = gbar.ravel(), 1.0 / sd.ravel() ** 2
    return np.array([((t * w) @ y) / ((t * w) @ t) for t in T])


def test_reference_parameters_are_stationary_for_the_oracle_gradient(lm_fixture, prep_fixture):
    """VERDICT r3 item 6.  At the parameters the REFERENCE fitted, the oracle's mean ELBO gradient (40 000 draws) is
      (a) per node, in units of the node's per-draw gradient noise, a Gaussian-looking cloud of width 0.03 -- the jitter
          ADAM leaves -- with no outliers;
      (b) uncorrelated with the expected pattern of each of the six additive gradient terms to within BETA_MAX;
      (c) and the test has POWER: scaling any one term INSIDE the oracle by 1 +- its documented threshold (3 % for four
          of them, 4 % and 7 % for the two the fixture's residual determines less well) breaks (b)."""
    from scipy import stats
    f, so, to = _fixture(lm_fixture, prep_fixture)
    n, p = f["n"], prep_fixture
    O.set_num_threads(1)  # (a 313-transcript sample: OpenMP only adds overhead)
    args = (so, to, f["effective_lengths"], p["mu"], p["omega"], p["alpha"])
    N, NP, seed = 40000, 3000, 1000
    gbar, se = O.vi_pin_stats(*args, seed, N)
    sd = se * np.sqrt(N)
    q = gbar / sd
    for b in range(3):
        assert 0.02 < q[b].std() < 0.045 and np.abs(q[b]).max() < 0.16 and abs(q[b].mean()) < 0.016, (b, q[b].std(), np.abs(q[b]).max(), q[b].mean())
        assert stats.shapiro(q[b]).pvalue > 1e-3
    T = term_patterns(so, to, f, p, 2000, seed)
    beta = term_betas(gbar, sd, T)
    print("beta", np.round(beta, 4))
    assert (np.abs(beta) < BETA_MAX).all(), beta
    # power: the same statistic on an oracle whose term j is biased (common random numbers: the first NP draws)
    base, _ = O.vi_pin_stats(*args, seed, NP)
    for j, (what, thr) in enumerate(TERMS):
        for sign in (-1.0, 1.0):
            O.set_debug_scale(j, 1.0 + sign * thr)
            try:
                biased, _ = O.vi_pin_stats(*args, seed, NP)
            finally:
                O.set_debug_scale(j, 1.0)
            bj = term_betas(gbar + (biased - base), sd, T)[j]
            assert abs(bj) > BETA_MAX[j], (what, sign * thr, bj)
    # ... and a parameter error is seen at once: mu moved by N(0, 0.1)
    rng = np.random.default_rng(0)
    moved, _ = O.vi_pin_stats(so, to, f["effective_lengths"], (p["mu"] + rng.normal(0, 0.1, n - 1)).astype(np.float32),
                              p["omega"], p["alpha"], seed, 4000)
    assert np.median(np.abs(moved[0])) > 4 * np.median(np.abs(gbar[0]))


def forward_chain(to, p, efflens, z0):
    """Julia-side forward path with its log-determinants (the oracle functions the fixtures pin), then the TF sampler's
    division by the effective lengths: returns log expression x and the three ladj values."""
    zs, ladj1 = O.sinh_asinh_transform(p["alpha"], z0, compute_ladj=True)
    ys, ladj2 = O.logit_normal_transform(p["mu"], np.exp(p["omega"]), zs, compute_ladj=True)
    q, ladj3 = to.transform(ys, compute_ladj=True)
    pe = q.astype(np.float64) / efflens
    pe /= pe.sum()
    return np.log(pe), ladj1, ladj2, ladj3


def test_density_of_a_sampler_draw_is_the_normal_density_minus_the_forward_log_determinants(lm_fixture, prep_fixture):
    f, so, to = _fixture(lm_fixture, prep_fixture)
    n, p, l = f["n"], prep_fixture, f["effective_lengths"].astype(np.float64)
    li, ri, fi = O.make_inverse_ptt_params(p["node_parent_idxs"], p["node_js"])
    sigma = np.exp(p["omega"])
    for seed in range(5):
        z0 = O.randn(n - 1, 4000 + seed)
        x, ladj1, ladj2, ladj3 = forward_chain(to, p, l, z0)
        # the TF sampler gives the same point (it clips at 1e-16 / 0.99999999: no draw of this fit comes near)
        xt = O.tf_sampler(z0, l, p["mu"], sigma, p["alpha"], li, ri, fi)[0]
        np.testing.assert_allclose(np.log(xt.astype(np.float64)), x, rtol=2e-5, atol=1e-5)  # (the TF side works in float32)
        # polee_approx_likelihood.py:384-400: sum x - (n-1) log sum e^x (= sum x: x is normalised), and the effective
        # length step  sum log l - log sum(p l)
        pe = np.exp(x)
        extra = x.sum() - (n - 1) * np.log(pe.sum()) + np.log(l).sum() - np.log((pe * l).sum())
        expect = (-np.log(2 * np.pi) * (n - 1) - (z0.astype(np.float64) ** 2).sum()) / 2 - (ladj1 + ladj2 + ladj3) + extra
        got = float(O.approx_log_prob(x.astype(np.float32), l, p["mu"], sigma, p["alpha"], li, ri, fi)[0])
        assert abs(got - expect) <= 1e-4 * abs(expect), (seed, got, expect)
