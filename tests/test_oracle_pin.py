"""Tighter pins of the CPU oracle on the reference's own fixtures.

(a) Stationarity: prep.h5 holds the parameters the REFERENCE fitted to likelihood-matrix.h5.  If the oracle's ELBO
    gradient is the reference's, those parameters are (nearly) stationary for it: the mean gradient over many draws is
    a small fraction of the individual terms it is made of.  A missing or biased term -- one of the ladj gradients of
    logitnormal.jl:50-53 / sinh_arcsinh.jl:35-36 / ptt.jl:197, the effective-length Jacobian of likelihood.jl:102-104,
    the likelihood gradient itself -- would leave a mean gradient of that term's size; a correlation test of fitted
    parameters cannot see that, this one can.
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


def mean_gradient(draw_gradients, n, ndraws, seed):
    acc = np.zeros((3, n - 1))
    for d in range(ndraws):
        acc += draw_gradients(O.randn(n - 1, seed + d))
    return acc / ndraws


def ladj_term_sizes(so, to, f, p, ndraws, seed):
    """median over nodes of |E[term]| for the ladj-gradient terms of mu, omega, alpha (in f64 from the oracle's ys)"""
    n = f["n"]
    mu, om, al = (p[k].astype(np.float64) for k in ("mu", "omega", "alpha"))
    sig = np.exp(om)
    acc = np.zeros((3, n - 1))
    for d in range(ndraws):
        z0 = O.randn(n - 1, seed + d)
        y = O.vi_draw_gradients(so, to, f["effective_lengths"], p["mu"], p["omega"], p["alpha"], z0)["ys"]
        c = al + np.arcsinh(z0.astype(np.float64))
        zs = np.sinh(c)
        acc[0] += 1 - 2 * y
        acc[1] += sig * (1 / sig + zs * (1 - 2 * y))
        acc[2] += np.cosh(c) * sig * (1 - 2 * y) + np.tanh(c)
    return np.median(np.abs(acc / ndraws), axis=1)


def test_reference_parameters_are_stationary_for_the_oracle_gradient(lm_fixture, prep_fixture):
    f, so, to = _fixture(lm_fixture, prep_fixture)
    n, p = f["n"], prep_fixture

    def grads_at(mu):
        def one(z0):
            r = O.vi_draw_gradients(so, to, f["effective_lengths"], mu, p["omega"], p["alpha"], z0)
            return np.stack([r["mu_grad"], r["omega_grad"], r["alpha_grad"]]).astype(np.float64)
        return one

    N = 20000
    gbar = mean_gradient(grads_at(p["mu"]), n, N, 1000)
    size = np.median(np.abs(gbar), axis=1)
    terms = ladj_term_sizes(so, to, f, p, 2000, 1000)
    # measured: |mean gradient| 0.03 .. 0.04 per block against ladj terms of 0.65 / 0.70 / 0.33 (the residual is the
    # reference's own ADAM noise after 500 steps)
    assert (size < 0.15 * terms).all(), (size, terms)
    # the statistic does respond: the same parameters with mu moved by N(0, 0.1) are far from stationary
    rng = np.random.default_rng(0)
    moved = mean_gradient(grads_at((p["mu"] + rng.normal(0, 0.1, n - 1)).astype(np.float32)), n, 4000, 1000)
    assert np.median(np.abs(moved[0])) > 4 * size[0]


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
