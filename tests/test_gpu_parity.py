"""GPU parity tests: the HIP path, called through the C ABI (polee_amd -> libpolee_hip.so),
against the CPU oracle on the same seeded inputs.  Run with `pytest -m gpu` on an MI355X.

Tolerances: the north star asks for 1e-4 relative on log-likelihood and posterior-mean
effects; integer/index structure is exact.  Tighter bounds are used where the arithmetic
allows (f64 tree, f32 sparse sums)."""
import os

import numpy as np
import pytest

from conftest import random_tree
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import polee_amd
    return polee_amd


@pytest.fixture(scope="module")
def ctx(P):
    return P.Context(0)


TREES = [("random", 313), ("spine", 40), ("balanced", 1000), ("random", 2), ("random", 3), ("spine", 3000),
         ("random", 5000)]


@pytest.mark.parametrize("kind,n", TREES)
def test_transform_matches_oracle(P, ctx, kind, n):
    rng = np.random.default_rng(n)
    p, js = random_tree(n, rng, kind)
    t = P.PolyaTreeTransform(p, js, ctx=ctx)
    to = O.PTT(p, js)
    lo, hi = (0.3, 0.7) if kind != "spine" or n < 100 else (0.02, 0.98)
    B = 3
    ys = rng.uniform(lo, hi, size=(B, n - 1))
    xs, ladj = t.transform(ys, compute_ladj=True)
    for b in range(B):
        xo, lo_ = to.transform(ys[b], True)
        np.testing.assert_allclose(xs[b], xo, rtol=3e-7, atol=0)
        if np.isfinite(lo_):
            assert abs(ladj[b] - lo_) <= 1e-10 * max(1.0, abs(lo_))
    assert xs.dtype == np.float32 and (xs >= np.float32(1e-16)).all()


@pytest.mark.parametrize("kind,n", [("random", 313), ("spine", 40), ("balanced", 1000), ("random", 5000)])
def test_transform_gradients_match_oracle(P, ctx, kind, n):
    rng = np.random.default_rng(n + 1)
    p, js = random_tree(n, rng, kind)
    t = P.PolyaTreeTransform(p, js, ctx=ctx)
    to = O.PTT(p, js)
    ys = rng.uniform(0.3, 0.7, size=(2, n - 1))
    xg = rng.normal(size=(2, n)) * 10
    yg = t.transform_gradients(ys, xg)
    yg0 = t.transform_gradients_no_ladj(ys, xg)
    for b in range(2):
        to.transform(ys[b], False)
        ref = to.transform_gradients(ys[b], xg[b])
        ref0 = to.transform_gradients_no_ladj(ys[b], xg[b])
        # the reference keeps f32 intermediates (ptt.jl:62): compare at f32 resolution of the
        # two cancelling terms
        us = to.us
        scale = np.abs(ref) + 1 + np.abs(xg[b]).max() * 64
        assert (np.abs(yg[b] - ref) <= 2e-6 * scale).all()
        assert (np.abs(yg0[b] - ref0) <= 2e-6 * scale).all()


def test_transform_gradients_f64_reference(P, ctx):
    """Tight check against an f64 restatement of ptt.jl:167-209 (no f32 intermediates)."""
    rng = np.random.default_rng(5)
    n = 400
    p, js = random_tree(n, rng, "random")
    t = P.PolyaTreeTransform(p, js, ctx=ctx)
    to = O.PTT(p, js)
    ys = rng.uniform(0.05, 0.95, n - 1)
    xg = rng.normal(size=n) * 100
    yg = t.transform_gradients(ys, xg)
    idx = to.index; N = to.N
    to.transform(ys, False); us = to.us
    g1 = np.zeros(N); g2 = np.zeros(N); ref = np.zeros(n - 1); k = n - 2
    for i in range(N - 1, -1, -1):
        if idx[0][i] > 0:
            g1[i] = xg[idx[0][i] - 1]
        else:
            l, r = idx[1][i] - 1, idx[2][i] - 1
            ref[k] = us[i] * ((g1[l] + g2[l]) - (g1[r] + g2[r]))
            g1[i] = ys[k] * g1[l] + (1 - ys[k]) * g1[r]
            g2[i] = 1 / us[i] + ys[k] * g2[l] + (1 - ys[k]) * g2[r]
            k -= 1
    np.testing.assert_allclose(yg, ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("kind,n", [("random", 313), ("spine", 30), ("balanced", 1000), ("random", 2)])
def test_inverse_and_tf_ops_match_oracle(P, ctx, kind, n):
    rng = np.random.default_rng(n + 2)
    p, js = random_tree(n, rng, kind)
    t = P.PolyaTreeTransform(p, js, ctx=ctx)
    to = O.PTT(p, js)
    l, r, f = P.make_inverse_ptt_params(p, js)
    lo_, ro_, fo_ = O.make_inverse_ptt_params(p, js)
    assert (l == lo_).all() and (r == ro_).all() and (f == fo_).all()
    B = 2
    x = rng.dirichlet(np.ones(n) * 0.5, size=B).astype(np.float32)
    x = np.maximum(x, np.float32(1e-12))
    ys, ladj = t.inverse_transform(x)
    for b in range(B):
        yo, lo2 = to.inverse_transform(x[b])
        np.testing.assert_allclose(ys[b], yo, rtol=1e-12)
        assert abs(ladj[b] - lo2) <= 1e-5 * max(1, abs(lo2))
    t2 = P.PolyaTreeTransform(ctx=ctx, index=(l, r, f))
    y_tf, ladj_tf = P.inv_hsb(x, t2)
    yo, lo3 = O.inv_hsb(x, l, r, f)
    np.testing.assert_allclose(y_tf, yo, rtol=1e-12)
    np.testing.assert_allclose(ladj_tf, lo3, rtol=2e-5, atol=1e-4)
    logit = rng.normal(0, 2, size=(B, n - 1)).astype(np.float32)
    np.testing.assert_allclose(P.hsb(logit, t2), O.hsb(logit, l, r, f), rtol=1e-6, atol=1e-38)
    yg = rng.normal(size=(B, n - 1))
    lg = rng.normal(size=B).astype(np.float32)
    bp = P.inv_hsb_grad(yg, lg, yo, t2)
    bo = O.inv_hsb_grad(yg, lg, yo, l, r, f)
    np.testing.assert_allclose(bp, bo, rtol=2e-6, atol=1e-6 * np.abs(bo).max())
    # round trip: inverse(transform(y)) == y
    y0 = rng.uniform(0.2, 0.8, n - 1)
    xr, _ = t.transform(y0)
    yr, _ = t.inverse_transform(xr)
    if kind != "spine":
        np.testing.assert_allclose(yr, y0, rtol=2e-5)


def _ygrad_f64(to, ys, x_grad):
    """f64 restatement of transform_gradients! (ptt.jl:167-209); also returns the magnitude of the
    two terms whose difference forms y_grad."""
    idx = to.index; N = to.N; n = to.n
    to.transform(ys, False); us = to.us
    g1 = np.zeros(N); g2 = np.zeros(N); out = np.zeros(n - 1); term = np.zeros(n - 1); k = n - 2
    for i in range(N - 1, -1, -1):
        if idx[0][i] > 0:
            g1[i] = x_grad[idx[0][i] - 1]
        else:
            l, r = idx[1][i] - 1, idx[2][i] - 1
            out[k] = us[i] * ((g1[l] + g2[l]) - (g1[r] + g2[r]))
            term[k] = us[i] * (abs(g1[l] + g2[l]) + abs(g1[r] + g2[r]))
            g1[i] = ys[k] * g1[l] + (1 - ys[k]) * g1[r]
            g2[i] = 1 / us[i] + ys[k] * g2[l] + (1 - ys[k]) * g2[r]
            k -= 1
    return out, term


def _gpu_sample(P, ctx, f, **kw):
    return P.RNASeqSample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"], f["effective_lengths"], ctx=ctx, **kw)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8])
def test_loglik_fixture_matches_oracle(P, ctx, lm_fixture, K):
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    rng = np.random.default_rng(K)
    x = rng.dirichlet(np.ones(f["n"]) * 0.3, size=K).astype(np.float32)
    x = np.clip(x, np.float32(1e-10), 1)
    lp, g = s.log_likelihood(x)
    lp0, g0 = s.log_likelihood(x, gradonly=True)
    assert (lp0 == 0).all()
    for k in range(K):
        lpo, go = so.log_likelihood(x[k])
        assert abs(lp[k] - lpo) <= 1e-6 * abs(lpo)
        np.testing.assert_allclose(g[k], go, rtol=2e-5, atol=1e-7 * np.abs(go).max())
        np.testing.assert_allclose(g0[k], go, rtol=2e-5, atol=1e-7 * np.abs(go).max())
        # homogeneity: sum_j x_j dlp/dx_j = m
        assert abs(float(g[k] @ x[k].astype(np.float64)) - f["m"]) < 1e-4 * f["m"]
    info = s.info
    assert info["nnz"] == 42775 and info["num_empty_rows"] == 0


def _random_matrix(rng, m, n, maxlen=6, empty_every=0, long_row=None):
    rows, cols = [], []
    for i in range(m):
        if empty_every and i % empty_every == 3:
            continue
        ln = 1 + int(rng.integers(0, maxlen))
        if long_row is not None and i == long_row[0]:
            ln = long_row[1]
        base = int(rng.integers(0, n))
        cs = np.unique((base + rng.integers(0, max(8, 2 * ln), size=ln)) % n)
        rows += [i] * cs.size; cols += cs.tolist()
    rows, cols = np.array(rows), np.array(cols)
    vals = np.exp(rng.normal(-9, 1.5, rows.size)).astype(np.float32)
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    colptr = np.zeros(n + 1, np.int64); np.add.at(colptr, cols + 1, 1)
    colptr = (np.cumsum(colptr) + 1).astype(np.uint32)
    return colptr, (rows + 1).astype(np.uint32), vals


def test_loglik_ragged_empty_rows_long_row_and_ks(P, ctx):
    rng = np.random.default_rng(11)
    m, n = 3000, 2500
    colptr, rowval, nzval = _random_matrix(rng, m, n, empty_every=40, long_row=(5, 700))
    ks = rng.integers(1, 6, m).astype(np.int64)
    x = np.clip(rng.dirichlet(np.ones(n), size=2), 1e-10, 1).astype(np.float32)
    s = P.RNASeqSample(m, n, colptr, rowval, nzval, ctx=ctx)
    sk = P.RNASeqSample(m, n, colptr.astype(np.uint64), rowval, nzval, ks=ks, ctx=ctx)
    assert s.info["num_empty_rows"] == 75 and s.info["max_row_nnz"] >= 500
    rows = rowval.astype(np.int64) - 1
    cols = np.repeat(np.arange(n), np.diff(colptr.astype(np.int64)))
    lp, g = s.log_likelihood(x)
    lpk, gk = P.factored_log_likelihood(sk, x)
    for k in range(2):
        sp = np.zeros(m); np.add.at(sp, rows, (nzval * x[k][cols]).astype(np.float64))
        live = sp > 0
        assert abs(lp[k] - np.log(sp[live]).sum()) <= 1e-6 * abs(lp[k])
        assert abs(lpk[k] - (ks[live] * np.log(sp[live])).sum()) <= 1e-6 * abs(lpk[k])
        gg = np.zeros(n); np.add.at(gg, cols, nzval.astype(np.float64) / sp[rows])
        ggk = np.zeros(n); np.add.at(ggk, cols, nzval.astype(np.float64) * ks[rows] / sp[rows])
        np.testing.assert_allclose(g[k], gg, rtol=3e-5, atol=1e-7 * gg.max())
        np.testing.assert_allclose(gk[k], ggk, rtol=3e-5, atol=1e-7 * ggk.max())
    # the 700-transcript fragments form their own oversized tiles (second small launch); all K up to 8 work there
    x8 = np.clip(rng.dirichlet(np.ones(n), size=8), 1e-10, 1).astype(np.float32)
    lp8, g8 = s.log_likelihood(x8)
    for k in (0, 7):
        sp = np.zeros(m); np.add.at(sp, rows, (nzval * x8[k][cols]).astype(np.float64))
        gg = np.zeros(n); np.add.at(gg, cols, nzval.astype(np.float64) / sp[rows])
        assert abs(lp8[k] - np.log(sp[sp > 0]).sum()) <= 1e-6 * abs(lp8[k])
        np.testing.assert_allclose(g8[k], gg, rtol=3e-5, atol=1e-7 * gg.max())
    # Xt entry point gives identical results
    so = O.Sample(m, n, colptr, rowval, nzval)
    s2 = P.RNASeqSample(m, n, None, None, None, ctx=ctx, xt=so.csr())
    lp2, g2 = s2.log_likelihood(x)
    np.testing.assert_allclose(lp2, lp, rtol=1e-12)
    np.testing.assert_allclose(g2, g, rtol=1e-6)


def _equivalence_class_matrix(rng, n=300, classes=40, copies=200):
    """Fragments in equivalence classes: `copies` rows per transcript set, as salmon's eq-classes / real data."""
    import scipy.sparse as sp
    rows, cols, vals = [], [], []
    r = 0
    for c in range(classes):
        w = int(rng.integers(1, 30))
        ts = np.sort(rng.choice(n, size=w, replace=False))
        for _ in range(copies + int(rng.integers(0, 70))):
            rows += [r] * w
            cols += ts.tolist()
            vals += rng.uniform(1e-6, 1e-3, w).tolist()
            r += 1
    X = sp.csc_matrix((np.array(vals, np.float32), (rows, cols)), shape=(r, n))
    X.sort_indices()
    return r, n, (X.indptr + 1).astype(np.uint32), (X.indices + 1).astype(np.uint32), X.data.astype(np.float32)


def test_factored_loglik_on_equivalence_classes(P, ctx):
    """factored_log_likelihood (likelihood.jl:59-85) where almost every slice is uniform: the multiplicities reach
    the matrix-core kernel through the slice stream."""
    rng = np.random.default_rng(12)
    m, n, colptr, rowval, nzval = _equivalence_class_matrix(rng)
    ks = rng.integers(1, 50, m).astype(np.int64)
    sk = P.RNASeqSample(m, n, colptr, rowval, nzval, ks=ks, ctx=ctx)
    assert sk.info["stream_nnz"][0] + sk.info["stream_nnz"][2] > 0.8 * sk.info["nnz"]
    x = np.clip(rng.dirichlet(np.ones(n), size=3), 1e-10, 1).astype(np.float32)
    lpk, gk = P.factored_log_likelihood(sk, x)
    rows = rowval.astype(np.int64) - 1
    cols = np.repeat(np.arange(n), np.diff(colptr.astype(np.int64)))
    for k in range(3):
        sp_ = np.zeros(m); np.add.at(sp_, rows, (nzval * x[k][cols]).astype(np.float64))
        assert abs(lpk[k] - (ks * np.log(sp_)).sum()) <= 1e-6 * abs(lpk[k])
        gg = np.zeros(n); np.add.at(gg, cols, nzval.astype(np.float64) * ks[rows] / sp_[rows])
        np.testing.assert_allclose(gk[k], gg, rtol=3e-5, atol=1e-7 * gg.max())


def test_loglik_rejects_bad_input(P, ctx):
    with pytest.raises(P.PoleeError):
        P.RNASeqSample(3, 2, np.array([1, 2, 3], np.uint32), np.array([1, 9], np.uint32), np.ones(2, np.float32), ctx=ctx)
    f_rows = np.ones(1100, np.uint32)  # one fragment compatible with 1100 transcripts: unsupported
    with pytest.raises(P.PoleeError):
        P.RNASeqSample(1, 1100, np.arange(1, 1102, dtype=np.uint32), f_rows, np.ones(1100, np.float32), ctx=ctx)
    with pytest.raises(P.PoleeError):
        P.PolyaTreeTransform(np.array([0, 1, 1, 2, 2], np.int32), np.array([0, 0, 1, 2, 2], np.int32), ctx=ctx)


def test_efflen_jacobian(P, ctx, lm_fixture):
    rng = np.random.default_rng(5)
    n = lm_fixture["n"]
    x = rng.dirichlet(np.ones(n), size=2).astype(np.float32)
    l = lm_fixture["effective_lengths"]
    g0 = rng.normal(size=(2, n))
    xls, g = P.effective_length_jacobian_adjustment(l, x, g0, ctx=ctx)
    for k in range(2):
        xo, go = O.effective_length_jacobian_adjustment(l, x[k], g0[k])
        np.testing.assert_allclose(g[k], go, rtol=1e-9)
        np.testing.assert_allclose(xls[k], xo, rtol=2e-6)


def test_gene_noninformative_prior(P, ctx, lm_fixture):
    """gene_noninformative_prior! (likelihood.jl:114-159) on top of the effective-length adjustment."""
    rng = np.random.default_rng(6)
    n = lm_fixture["n"]
    x = rng.dirichlet(np.ones(n), size=3).astype(np.float32)
    l = lm_fixture["effective_lengths"]
    g0 = rng.normal(size=(3, n))
    gene_of = rng.integers(0, n // 3, size=n).astype(np.int32)
    gene_of[rng.random(n) < 0.1] = -1  # transcripts without a known gene
    xls, g1 = P.effective_length_jacobian_adjustment(l, x, g0, ctx=ctx)
    g2 = P.gene_noninformative_prior(l, xls, x, g1, gene_of, ctx=ctx)
    for k in range(3):
        go = O.gene_noninformative_prior(l, xls[k], x[k], g1[k], gene_of)
        np.testing.assert_allclose(g2[k], go, rtol=1e-9, atol=1e-9 * np.abs(go).max())
        assert np.abs(g2[k] - g1[k]).max() > 0  # the prior does contribute
    # the reference's Dict form (1-based transcript indexes), single vector
    d = {}
    for i, gi in enumerate(gene_of):
        if gi >= 0:
            d.setdefault("g%d" % gi, []).append(i + 1)
    g3 = P.gene_noninformative_prior(l, xls[0], x[0], g1[0], d, ctx=ctx)
    np.testing.assert_allclose(g3, g2[0], rtol=1e-12)
    # no gene information at all: unchanged
    g4 = P.gene_noninformative_prior(l, xls[0], x[0], g1[0], np.full(n, -1, np.int32), ctx=ctx)
    np.testing.assert_array_equal(g4, g1[0])


@pytest.mark.parametrize("use_efflen", [True, False])
def test_vi_single_step_gradients_match_oracle(P, ctx, lm_fixture, prep_fixture, use_efflen):
    """One VI iteration's K draws at the reference's own fitted parameters, device RNG noise
    exported and fed to the oracle."""
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    K = 6
    fit = P.LikelihoodApproximationFit(s, t, num_steps=2, num_mc_samples=K, use_efflen_jacobian=use_efflen,
                                       gradonly=False, seed=99)
    fit.set_params(prep_fixture["mu"], prep_fixture["omega"], prep_fixture["alpha"])
    z0 = fit.export_noise(1)
    assert abs(z0.mean()) < 0.1 and abs(z0.std() - 1) < 0.05
    out = fit.eval_gradients()
    mu_g = np.zeros(f["n"] - 1); om_g = np.zeros_like(mu_g); al_g = np.zeros_like(mu_g)
    mu_o = np.zeros_like(mu_g); om_o = np.zeros_like(mu_g); al_o = np.zeros_like(mu_g); noise = np.zeros_like(mu_g)
    mu, omega, alpha = (prep_fixture[k].astype(np.float64) for k in ("mu", "omega", "alpha"))
    sigma = np.exp(omega)
    for d in range(K):
        r = O.vi_draw_gradients(so, to, f["effective_lengths"], prep_fixture["mu"], prep_fixture["omega"],
                                prep_fixture["alpha"], z0[d], use_efflen_jacobian=use_efflen)
        np.testing.assert_allclose(out["xs"][d], r["xs"], rtol=1e-5)
        assert abs(out["lp"][d] - r["lp"]) <= 1e-6 * abs(r["lp"])
        assert abs(out["ladj"][d] - r["ladj"]) <= 1e-5 * max(1, abs(r["ladj"]))
        np.testing.assert_allclose(out["x_grad"][d], r["x_grad"], rtol=1e-4, atol=1e-6 * np.abs(r["x_grad"]).max())
        # y_grad: tight against an f64 restatement of ptt.jl:167-209 fed with the oracle's own
        # x_grad; the reference itself keeps f32 intermediates (ptt.jl:62), so against the oracle the
        # bound is f32 rounding of the two cancelling terms u_i*(g1+g2)_left and u_i*(g1+g2)_right.
        yg64, term = _ygrad_f64(to, r["ys"], r["x_grad"])
        assert (np.abs(out["y_grad"][d] - yg64) <= 2e-5 * (term + 1)).all()
        assert (np.abs(r["y_grad"] - yg64) <= 2e-5 * (term + 1)).all()
        # chain rule through logit-normal (logitnormal.jl:38-55) and sinh-arcsinh (sinh_arcsinh.jl:29-38)
        # in f64 from yg64: the reference for the GPU's per-step gradients
        y = r["ys"]; c = alpha + np.arcsinh(z0[d].astype(np.float64)); zs = np.sinh(c); dyy = y * (1 - y)
        mu_g += dyy * yg64 + (1 - 2 * y)
        sg = dyy * zs * yg64 + 1 / sigma + zs * (1 - 2 * y)
        zg = dyy * sigma * yg64 + sigma * (1 - 2 * y)
        om_g += sigma * sg
        al_g += np.cosh(c) * zg + np.tanh(c)
        noise += dyy * term
        mu_o += r["mu_grad"]; om_o += r["omega_grad"]; al_o += r["alpha_grad"]
    for name, acc, orc, fac in (("mu_grad", mu_g, mu_o, 1.0), ("omega_grad", om_g, om_o, np.abs(sigma) * 3 + 1),
                                ("alpha_grad", al_g, al_o, np.abs(sigma) * 30 + 1)):
        ref = acc / K
        assert (np.abs(out[name] - ref) <= 1e-4 * np.abs(ref) + 1e-5 * fac * (noise / K + 1)).all(), name
        # the oracle (f32 intermediates, as the reference) agrees within its own rounding noise
        assert (np.abs(orc / K - ref) <= 1e-4 * np.abs(ref) + 1e-4 * fac * (noise / K + 1)).all(), name


@pytest.mark.parametrize("kind,K", [("spine", 6), ("random", 6), ("balanced", 8), ("spine", 8), ("random", 3)])
def test_vi_subtree_sums_across_chunks_and_sixteen_decades(P, ctx, kind, K):
    """The VI loop's backward pass (round 6: a chunk-local double-double prefix per 512 leaves -- 256 for K > 6 --, the nodes inside a
    chunk from LDS rows, the chunk-crossing ones through exported rows + chunk offsets, subtree sums kept as Float32) on trees of
    3 000 leaves, i.e. several chunks: a caterpillar (every node crosses chunks), a random and a balanced tree; parameters spread
    so that the leaves' u spans more than sixteen decades (hard part 2: tiny subtrees beside large ones).  y_grad of every
    draw against the f64 restatement of ptt.jl:167-209 fed with the oracle's x gradient, in the reference's own rounding class
    (its gradient intermediates are Float32: 2e-5 of the two cancelling terms), and the three parameter gradients."""
    from tools import synth
    n, m = 3000, 60000
    smp = synth.make_sample(n, m, 4.0, 17)
    colptr, rowval, nzval = synth.to_csc(smp)
    eff = smp["effective_lengths"]
    rng = np.random.default_rng(5)
    parents, js = random_tree(n, rng, kind)
    s = P.RNASeqSample(m, n, colptr, rowval, nzval, eff, ctx=ctx)
    t = P.PolyaTreeTransform(parents, js, ctx=ctx)
    so, to = O.Sample(m, n, colptr, rowval, nzval), O.PTT(parents, js)
    mu = rng.normal(0, 3.0, n - 1).astype(np.float32)
    if kind == "spine":
        # a caterpillar multiplies n - 1 factors along its spine: in f64 the reference's own recursion underflows (1 / u = inf) unless
        # the factor towards the larger subtree stays near 1 -- so y ~ 0.9975 towards it (the chain's u decays to 5e-4), noise on
        # top, and forty extreme nodes for tiny subtrees
        import ctypes as C
        from polee_amd import _lib as L
        N = 2 * n - 1
        code = np.zeros(3 * n - 2, np.uint32); tgt = np.zeros(3 * n - 2, np.int32); ltid = np.zeros(n, np.int32)
        lo, mid, hi1 = (np.zeros(n - 1, np.int32) for _ in range(3))
        depth = C.c_int32()
        pp, jj = np.ascontiguousarray(parents, np.int32), np.ascontiguousarray(js, np.int32)
        L.check(L.lib().polee_debug_ptt_plan(L.ptr(pp, L.i32p), L.ptr(jj, L.i32p), N, L.ptr(code, L.u32p), L.ptr(tgt, L.i32p),
                                             L.ptr(ltid, L.i32p), L.ptr(lo, L.i32p), L.ptr(mid, L.i32p), L.ptr(hi1, L.i32p), C.byref(depth)))
        left_bigger = (hi1 - mid) > (mid - lo)  # y multiplies the LEFT child (ptt.jl:147)
        mu = (np.where(left_bigger, 6.0, -6.0) + rng.normal(0, 0.5, n - 1)).astype(np.float32)
        mu[rng.integers(0, n - 1, 40)] = rng.choice([-18.0, 18.0], 40).astype(np.float32)
    omega = rng.normal(-1.5, 0.3, n - 1).astype(np.float32)
    alpha = rng.normal(0, 0.2, n - 1).astype(np.float32)
    fit = P.LikelihoodApproximationFit(s, t, num_steps=2, num_mc_samples=K, gradonly=False, seed=7)
    fit.set_params(mu, omega, alpha)
    z0 = fit.export_noise(1)
    out = fit.eval_gradients()
    span = []
    mu_acc, noise = np.zeros(n - 1), np.zeros(n - 1)
    for d in range(K):
        r = O.vi_draw_gradients(so, to, eff, mu, omega, alpha, z0[d])
        np.testing.assert_allclose(out["xs"][d], r["xs"], rtol=2e-5)
        np.testing.assert_allclose(out["x_grad"][d], r["x_grad"], rtol=1e-4, atol=1e-6 * np.abs(r["x_grad"]).max())
        yg64, term = _ygrad_f64(to, r["ys"], r["x_grad"])
        bad = np.abs(out["y_grad"][d] - yg64) > 2e-5 * (term + 1)
        assert not bad.any(), (kind, K, d, int(bad.sum()), float((np.abs(out["y_grad"][d] - yg64) / (term + 1)).max()))
        span.append((r["xs"].min(), r["xs"].max()))
        dyy = r["ys"] * (1 - r["ys"])
        mu_acc += dyy * yg64 + (1 - 2 * r["ys"])
        noise += dyy * term
    assert min(a for a, _ in span) <= 1.001e-10 and max(b for _, b in span) > 1e-3, span  # (xs is clamped at 1e-10: u goes far below)
    # mu's gradient, the K-draw mean of dyy y_grad + (1 - 2 y) (logitnormal.jl:50-52), against the same chain in f64 from yg64 (the
    # oracle's own Float32 recursion overflows on the caterpillar's extreme nodes: 1 / u is not a Float32)
    ref = mu_acc / K
    ok = np.abs(out["mu_grad"] - ref) <= 1e-4 * np.abs(ref) + 1e-5 * (noise / K + 1)
    assert ok.all(), (kind, K, int((~ok).sum()))


def test_vi_trajectory_with_supplied_noise(P, ctx, lm_fixture, prep_fixture):
    """Five full iterations (sampling, likelihood, backward, ADAM) with identical z0 on both sides."""
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    steps, K = 5, 6
    z0 = O.randn(steps * K * (f["n"] - 1), 3)
    ref = O.approximate_likelihood(so, to, f["effective_lengths"], num_steps=steps, num_mc=K, z0=z0, gradonly=False)
    # initial values first
    fit0 = P.LikelihoodApproximationFit(s, t, num_steps=steps, num_mc_samples=K, z0=z0)
    init = O.approximate_likelihood(so, to, f["effective_lengths"], init_only=True)
    mu0, om0, al0 = fit0.params()
    np.testing.assert_allclose(mu0, init["mu"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(om0, init["omega"]); np.testing.assert_array_equal(al0, init["alpha"])
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=steps, num_mc_samples=K, z0=z0,
                                   gradonly=False)
    # ADAM's first steps move every parameter by +-max_step whatever the gradient's size, so a
    # parameter whose gradient is ~0 is sign-sensitive to rounding: require near-all to agree.
    for key, tol in (("mu", 2e-4), ("omega", 2e-4), ("alpha", 2e-4)):
        ok = np.abs(got[key] - ref[key]) <= tol * (1 + np.abs(ref[key]))
        assert ok.mean() >= 0.99, (key, ok.mean())
    np.testing.assert_allclose(got["lp_mean"], ref["lp_mean"], rtol=1e-5)
    np.testing.assert_allclose(got["elbo"], ref["elbo"], rtol=1e-5)
    assert (got["node_parent_idxs"] == prep_fixture["node_parent_idxs"]).all()


def test_vi_trajectory_with_gene_noninformative_prior(P, ctx, lm_fixture, prep_fixture):
    """gene_noninformative = true as an option of the loop (likelihood-approximation.jl:475-491, 535-538): single-step
    x gradients and a five-iteration trajectory against the oracle with the same gene annotation and noise."""
    f = lm_fixture
    n = f["n"]
    s = _gpu_sample(P, ctx, f)
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    so = O.Sample(f["m"], n, f["colptr"], f["rowval"], f["nzval"])
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    rng = np.random.default_rng(21)
    gene_of = rng.integers(0, n // 3, size=n).astype(np.int32)
    gene_of[rng.random(n) < 0.1] = -1  # transcripts without a known gene
    steps, K = 5, 6
    z0 = O.randn(steps * K * (n - 1), 9)
    ref = O.approximate_likelihood(so, to, f["effective_lengths"], num_steps=steps, num_mc=K, z0=z0, gradonly=False,
                                   gene_of=gene_of)
    plain = O.approximate_likelihood(so, to, f["effective_lengths"], num_steps=steps, num_mc=K, z0=z0, gradonly=False)
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=steps, num_mc_samples=K, z0=z0,
                                   gradonly=False, gene_noninformative=True, gene_transcripts=gene_of)
    assert np.abs(ref["mu"] - plain["mu"]).max() > 1e-2  # the prior does change the trajectory
    for key in ("mu", "omega", "alpha"):
        ok = np.abs(got[key] - ref[key]) <= 2e-4 * (1 + np.abs(ref[key]))
        assert ok.mean() >= 0.99, (key, ok.mean())
    np.testing.assert_allclose(got["lp_mean"], ref["lp_mean"], rtol=1e-5)
    # the Dict form of the reference (gene id -> 1-based transcript indexes) gives the same fit
    d = {}
    for i, gi in enumerate(gene_of):
        if gi >= 0:
            d.setdefault("g%d" % gi, []).append(i + 1)
    got2 = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=steps, num_mc_samples=K, z0=z0,
                                    gene_noninformative=True, gene_transcripts=d)
    np.testing.assert_allclose(got2["mu"], got["mu"], rtol=1e-4, atol=1e-4)
    # single-draw x gradients (the hook returns x_grad after both adjustments)
    fit = P.LikelihoodApproximationFit(s, t, num_steps=2, num_mc_samples=K, gradonly=False, seed=5, gene_transcripts=gene_of)
    fit.set_params(prep_fixture["mu"], prep_fixture["omega"], prep_fixture["alpha"])
    zz = fit.export_noise(1)
    out = fit.eval_gradients()
    for dd_ in range(K):
        r = O.vi_draw_gradients(so, to, f["effective_lengths"], prep_fixture["mu"], prep_fixture["omega"],
                                prep_fixture["alpha"], zz[dd_])
        xs = r["xs"]
        xl = (xs / f["effective_lengths"]).astype(np.float32)
        xls = (xl.astype(np.float64) / xl.astype(np.float64).sum()).astype(np.float32)
        want = O.gene_noninformative_prior(f["effective_lengths"], xls, xs, r["x_grad"], gene_of)
        np.testing.assert_allclose(out["x_grad"][dd_], want, rtol=1e-4, atol=1e-6 * np.abs(want).max())
    # no gene information at all: warning, option off (likelihood-approximation.jl:487-490)
    with pytest.warns(UserWarning, match="no gene information"):
        got3 = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=steps, num_mc_samples=K, z0=z0,
                                        gradonly=False, gene_noninformative=True)
    np.testing.assert_allclose(got3["lp_mean"], plain["lp_mean"], rtol=1e-5)


def test_deterministic_mode_is_bitwise_reproducible(P, ctx, lm_fixture, prep_fixture):
    """polee_loglik_set_deterministic: fixed-order sums instead of float atomics.  Two evaluations of the same inputs
    and two whole fits with the same noise agree BIT FOR BIT (the default mode does not promise that), and the values
    are the default mode's up to float32 summation order."""
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    rng = np.random.default_rng(3)
    x = rng.dirichlet(np.ones(f["n"]), size=6).astype(np.float32)
    lp0, g0 = s.log_likelihood(x)
    s.set_deterministic(True)
    lp1, g1 = s.log_likelihood(x)
    for _ in range(3):
        lp2, g2 = s.log_likelihood(x)
        assert np.array_equal(lp1, lp2) and np.array_equal(g1, g2)
    # (lp: a lane keeps the float32 PRODUCT of its row sums' mantissas, so the value depends on which slices a lane saw -- the
    # static and the dynamic schedule cut the tiles differently: ~1e-11 relative)
    np.testing.assert_allclose(lp1, lp0, rtol=1e-9)
    np.testing.assert_allclose(g1, g0, rtol=2e-5, atol=1e-6 * np.abs(g0).max())
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    for k in range(6):
        lpo, go = so.log_likelihood(x[k])
        assert abs(lp1[k] - lpo) <= 1e-6 * abs(lpo)
        np.testing.assert_allclose(g1[k], go, rtol=1e-4, atol=1e-6 * np.abs(go).max())
    s.set_deterministic(False)
    steps, K = 40, 6
    z0 = O.randn(steps * K * (f["n"] - 1), 31)
    fits = []
    for _ in range(2):
        fit = P.LikelihoodApproximationFit(s, t, num_steps=steps, num_mc_samples=K, z0=z0, gradonly=False, deterministic=True)
        fit.run(steps)
        fit.sync()
        fits.append((fit.params(), fit.trace()))
    for a, b in zip(fits[0][0], fits[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(fits[0][1][1], fits[1][1][1])  # the E[lp] trace too
    s.set_deterministic(False)


def test_deterministic_gradients_do_not_depend_on_the_tile_schedule():
    """Round 5: the deterministic mode draws its tiles from the global counter like the default (only lp keeps the static lists).  A
    tile's sums go to the TILE's slot in wave order and a transcript's slots are added in tile order, so which workgroup took
    which tile must not show: gradient-only passes over a sample of ~3 000 tiles, repeated (the draws differ from launch to
    launch), give the same bits, and the same bits as a process that runs the static lists (POLEE_DET_STATIC=1).  The sample has
    699 rows of more than 32 transcripts (stream B, the per-tile kernel): until this test they added with float atomics and one
    gradient entry moved by an ulp between launches, whatever the schedule; in this mode the kernel now gives a tile to one wave and
    stores its sums to the tile's slots (psell_tile_body).  Passes with lp are compared too."""
    import hashlib
    import subprocess
    import sys
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import polee_amd as P
from tools import synth
smp = synth.make_sample(20000, 3000000, 8.0, 77, literal=True)
c, r, v = synth.to_csc(smp)
ctx = P.Context(0)
s = P.RNASeqSample(smp["m"], smp["n"], c, r, v, ctx=ctx)
assert s.info["stream_rows"][5] > 0  # (rows of more than 32 transcripts, stream B: one wave per tile and the tile's slots in this mode)
assert s.info["stream_rows"][6] == 0  # (stream C -- rows without any structure -- adds with float atomics: outside the guarantee)
s.set_deterministic(True)
x = np.random.default_rng(5).dirichlet(np.ones(smp["n"]), size=6).astype(np.float32)
hs = set()
for _ in range(6):
    lp, g = s.log_likelihood(x, gradonly=True)
    hs.add(hashlib.sha256(np.ascontiguousarray(g).tobytes()).hexdigest())
assert len(hs) == 1, hs
hl = set()
for _ in range(4):  # with lp (the static lists): value and gradient, bit for bit
    lp, g = s.log_likelihood(x)
    hl.add(hashlib.sha256(np.ascontiguousarray(g).tobytes() + np.ascontiguousarray(lp).tobytes()).hexdigest())
assert len(hl) == 1, hl
print("HASH", hs.pop(), s.info["num_tiles"])
""" % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."),)
    out = []
    for static in (False, True):
        env = dict(os.environ)
        env.pop("POLEE_DET_STATIC", None)
        if static:
            env["POLEE_DET_STATIC"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0].split()
        assert int(line[2]) > 900  # (enough tiles for the schedules to differ)
        out.append(line[1])
    assert out[0] == out[1], out


def _expected_loglik(so, to, mu, omega, alpha, efflens, ndraws, seed):
    sigma = np.exp(omega)
    lps, pm = [], np.zeros(to.n)
    for d in range(ndraws):
        z0 = O.randn(to.n - 1, seed + d)
        x = O.sampler_draw(to, mu, sigma, alpha, z0)
        lps.append(so.log_likelihood(np.clip(x, np.float32(1e-10), np.float32(1.0)))[0])
        xe = x.astype(np.float64) / efflens
        pm += xe / xe.sum()
    return np.array(lps), pm / ndraws


def test_full_fit_matches_reference_fit_statistically(P, ctx, lm_fixture, prep_fixture):
    """500 x 6 production fit (device RNG, gradonly) vs the reference's own fit of the same X
    (prep.h5): expected log-likelihood within MC error, posterior means within 1e-4-level
    agreement of their log-correlation."""
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t)
    l = f["effective_lengths"]
    lp_ref, pm_ref = _expected_loglik(so, to, prep_fixture["mu"], prep_fixture["omega"], prep_fixture["alpha"], l, 200, 1000)
    lp_fit, pm_fit = _expected_loglik(so, to, got["mu"], got["omega"], got["alpha"], l, 200, 5000)
    assert abs(lp_fit.mean() - lp_ref.mean()) < 6 * np.hypot(lp_ref.std(), lp_fit.std()) / np.sqrt(200) + 5
    expressed = pm_ref > 1e-4
    r = np.corrcoef(np.log(pm_ref[expressed]), np.log(pm_fit[expressed]))[0, 1]
    assert r > 0.99, r
    # node by node against the reference's OWN fitted parameters (same tree, different RNG): the agreement is at
    # the level of our own seed-to-seed variation (measured: r = 0.9994 / 0.9995 / 0.979 vs 0.9995 / 0.9995 / 0.981)
    other = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, seed=2)
    for key, rmin in (("mu", 0.998), ("omega", 0.998), ("alpha", 0.95)):
        r_ref = np.corrcoef(got[key], prep_fixture[key])[0, 1]
        r_self = np.corrcoef(got[key], other[key])[0, 1]
        assert r_ref > rmin and r_ref > r_self - 0.01, (key, r_ref, r_self)
        d_ref = np.median(np.abs(got[key] - prep_fixture[key]))
        d_self = np.median(np.abs(got[key] - other[key]))
        assert d_ref < 2.0 * d_self + 1e-3, (key, d_ref, d_self)


def test_fit_with_own_tree_is_as_good_as_the_reference_fit(P, ctx, lm_fixture, prep_fixture):
    """approximate_likelihood(approx, sample) with NO tree given: the tree comes from polee_hclust
    (PolyaTreeTransform(X, :cluster), ptt.jl:35-52).  The fitted approximation must explain the data as well as
    the reference's own fit (which used its stored tree): expected log-likelihood within MC error + a small
    margin, posterior means strongly correlated."""
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox("cluster"), s)
    assert len(got["node_js"]) == 2 * f["n"] - 1
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    to_ref = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    to_own = O.PTT(got["node_parent_idxs"], got["node_js"])
    l = f["effective_lengths"]
    lp_ref, pm_ref = _expected_loglik(so, to_ref, prep_fixture["mu"], prep_fixture["omega"], prep_fixture["alpha"], l, 100, 1000)
    lp_fit, pm_fit = _expected_loglik(so, to_own, got["mu"], got["omega"], got["alpha"], l, 100, 7000)
    assert lp_fit.mean() > lp_ref.mean() - (6 * np.hypot(lp_ref.std(), lp_fit.std()) / np.sqrt(100) + 10)
    expressed = pm_ref > 1e-4
    r = np.corrcoef(np.log(pm_ref[expressed]), np.log(pm_fit[expressed]))[0, 1]
    assert r > 0.98, r
    # the sequential (list) tree is accepted as well
    got2 = P.approximate_likelihood(P.LogitSkewNormalPTTApprox("sequential"), s, num_steps=20)
    assert np.isfinite(got2["mu"]).all()


def test_sampler_matches_oracle(P, ctx, prep_fixture):
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    mu, sigma, alpha = prep_fixture["mu"], np.exp(prep_fixture["omega"]), prep_fixture["alpha"]
    als = P.ApproxLikelihoodSampler()
    als.set_transform(t, mu, sigma, alpha)
    z0 = np.stack([O.randn(t.n - 1, 50 + d) for d in range(11)])
    xs = als.rand(11, z0=z0)
    for d in range(11):
        np.testing.assert_allclose(xs[d], O.sampler_draw(to, mu, sigma, alpha, z0[d]), rtol=2e-5, atol=1e-30)
    # device RNG: draws are on the simplex and differ between calls
    a, b = als.rand(4), als.rand(4)
    assert np.allclose(a.sum(axis=1), 1, atol=1e-4) and not np.allclose(a, b)


def test_sampler_quantiles_and_posterior_mean(P, ctx, prep_fixture):
    """Statistics.quantile / posterior_mean over sampler draws (src/approx-sampler.jl:50-117) with supplied noise,
    against the oracle's draws and NumPy (np.quantile's default is Julia's default definition)."""
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    mu, sigma, alpha = prep_fixture["mu"], np.exp(prep_fixture["omega"]), prep_fixture["alpha"]
    als = P.ApproxLikelihoodSampler()
    als.set_transform(t, mu, sigma, alpha)
    for N in (1, 7, 100):
        z0 = np.stack([O.randn(t.n - 1, 900 + d) for d in range(N)])
        draws = np.stack([O.sampler_draw(to, mu, sigma, alpha, z0[d]) for d in range(N)])
        qs = (0.01, 0.25, 0.5, 0.99, 1.0)
        q = als.quantile(qs, N, z0=z0)
        qo = np.quantile(draws.astype(np.float64), qs, axis=0)
        np.testing.assert_allclose(q, qo, rtol=3e-5, atol=1e-30)
        pm = als.posterior_mean(N, z0=z0)
        acc = np.zeros(t.n, np.float32)
        for d in range(N):  # f32 accumulation in draw order, as the reference
            acc += np.clip(draws[d], np.float32(1e-15), np.float32(0.9999999))
        np.testing.assert_allclose(pm, acc / np.float32(N), rtol=3e-5, atol=1e-30)
    # more draws than fit in LDS (global fallback) and the device RNG: quantiles are ordered and bracket the mean draw
    q = als.quantile((0.05, 0.5, 0.95), 300)
    assert np.all(q[0] <= q[1]) and np.all(q[1] <= q[2])
    pm = als.posterior_mean(300)
    assert abs(pm.sum() - 1) < 1e-3
    with pytest.raises(P.PoleeError):
        als.quantile((1.5,), 10)


@pytest.mark.parametrize("shared", [False, True])
def test_approx_logprob_and_gradient_match_oracle(P, ctx, shared):
    rng = np.random.default_rng(21)
    n, S = 600, 3
    trees = [random_tree(n, rng) for _ in range(1 if shared else S)]
    idx = [O.make_inverse_ptt_params(*tr) for tr in trees]
    L_, R_, F_ = (np.stack([i[j] for i in idx]) for j in range(3))
    x = rng.normal(0, 1.5, size=(S, n)).astype(np.float32)
    eff = rng.uniform(200, 3000, size=(S, n)).astype(np.float32)
    mu = rng.normal(0, 1, size=(S, n - 1)).astype(np.float32)
    sigma = np.exp(rng.normal(-1, 0.3, size=(S, n - 1))).astype(np.float32)
    alpha = rng.normal(0, 0.3, size=(S, n - 1)).astype(np.float32)
    ap = P.RNASeqApproxLikelihood(dict(efflen=eff, la_mu=mu, la_sigma=sigma, la_alpha=alpha, left_index=L_,
                                       right_index=R_, leaf_index=F_), ctx=ctx)
    lp, g = ap.log_prob(x, want_grad=True)
    lpo, go = O.approx_log_prob(x, eff, mu, sigma, alpha, L_, R_, F_, want_grad=True)
    np.testing.assert_allclose(lp, lpo, rtol=1e-4)
    np.testing.assert_allclose(g, go, rtol=2e-3, atol=2e-3 * np.abs(go).max())
    np.testing.assert_allclose(ap.log_prob(x), lp, rtol=1e-6)
    z0 = rng.normal(size=(S, n - 1)).astype(np.float32)
    xs = ap.sample(z0)
    xo = O.tf_sampler(z0, eff, mu, sigma, alpha, L_, R_, F_)
    np.testing.assert_allclose(xs, xo, rtol=1e-4, atol=1e-16)


def test_gene_level_logprob_and_gradients_match_oracle(P, ctx):
    """RNASeqGeneApproxLikelihoodDist (polee_gene_expression.py:14-90) around the transcript density."""
    rng = np.random.default_rng(22)
    n, S, G = 500, 2, 140
    trees = [random_tree(n, rng) for _ in range(S)]
    idx = [O.make_inverse_ptt_params(*tr) for tr in trees]
    L_, R_, F_ = (np.stack([i[j] for i in idx]) for j in range(3))
    eff = rng.uniform(200, 3000, size=(S, n)).astype(np.float32)
    mu = rng.normal(0, 1, size=(S, n - 1)).astype(np.float32)
    sigma = np.exp(rng.normal(-1, 0.3, size=(S, n - 1))).astype(np.float32)
    alpha = rng.normal(0, 0.3, size=(S, n - 1)).astype(np.float32)
    gene_of = np.concatenate([np.arange(G), rng.integers(0, G, n - G)]).astype(np.int32)  # every gene non-empty
    rng.shuffle(gene_of)
    x_gene = rng.normal(0, 1.5, size=(S, G)).astype(np.float32)
    x_iso = rng.normal(0, 1.0, size=(S, n)).astype(np.float32)
    ap = P.RNASeqApproxLikelihood(dict(efflen=eff, la_mu=mu, la_sigma=sigma, la_alpha=alpha, left_index=L_,
                                       right_index=R_, leaf_index=F_), ctx=ctx)
    lp, gg, gi = ap.gene_log_prob(x_gene, x_iso, gene_of + 1, want_grad=True)
    lpo, ggo, gio = O.approx_gene_log_prob(x_gene, x_iso, gene_of, eff, mu, sigma, alpha, L_, R_, F_, want_grad=True)
    np.testing.assert_allclose(lp, lpo, rtol=1e-4)
    np.testing.assert_allclose(gg, ggo, rtol=3e-3, atol=3e-3 * np.abs(ggo).max())
    np.testing.assert_allclose(gi, gio, rtol=3e-3, atol=3e-3 * np.abs(gio).max())
    np.testing.assert_allclose(ap.gene_log_prob(x_gene, x_iso, gene_of + 1), lp, rtol=1e-6)
    # the within-gene parametrisation is shift invariant: adding a constant to a gene's isoforms changes nothing
    x_iso2 = x_iso + rng.normal(size=(S, G)).astype(np.float32)[:, gene_of]
    np.testing.assert_allclose(ap.gene_log_prob(x_gene, x_iso2, gene_of + 1), lp, rtol=2e-5)
    with pytest.raises(P.PoleeError):
        ap.gene_log_prob(x_gene, x_iso, np.full(n, G + 5, np.int32))


def test_feature_moments_match_sampler_draws(P, ctx):
    """approximate_feature_likelihood (polee_gene_expression.py:191-222) with supplied noise against the oracle's
    sampler (tf_sampler) and NumPy moments; then with the device RNG the two estimates agree within MC error."""
    rng = np.random.default_rng(23)
    n, S, F, D1, D2 = 300, 2, 40, 6, 5
    trees = [random_tree(n, rng) for _ in range(S)]
    idx = [O.make_inverse_ptt_params(*tr) for tr in trees]
    L_, R_, F_ = (np.stack([i[j] for i in idx]) for j in range(3))
    eff = rng.uniform(200, 3000, size=(S, n)).astype(np.float32)
    mu = rng.normal(0, 1, size=(S, n - 1)).astype(np.float32)
    sigma = np.exp(rng.normal(-1, 0.3, size=(S, n - 1))).astype(np.float32)
    alpha = rng.normal(0, 0.3, size=(S, n - 1)).astype(np.float32)
    ap = P.RNASeqApproxLikelihood(dict(efflen=eff, la_mu=mu, la_sigma=sigma, la_alpha=alpha, left_index=L_,
                                       right_index=R_, leaf_index=F_), ctx=ctx)
    gene_of = np.concatenate([np.arange(F), rng.integers(0, F, n - F)])
    fi, ti = gene_of + 1, np.arange(1, n + 1)
    z0 = rng.normal(size=(D1 + D2, S, n - 1)).astype(np.float32)
    loc, scale = ap.approximate_feature_likelihood(F, fi, ti, D1, D2, z0=z0)
    logs = []
    for d in range(D1 + D2):
        x = O.tf_sampler(z0[d], eff, mu, sigma, alpha, L_, R_, F_).astype(np.float64)
        fx = np.stack([np.bincount(gene_of, x[s_], F) for s_ in range(S)])
        logs.append(np.log(fx))
    logs = np.array(logs)
    loc_o = logs[:D1].mean(axis=0)
    scale_o = np.sqrt(((loc_o[None] - logs[D1:]) ** 2).mean(axis=0))
    np.testing.assert_allclose(loc, loc_o, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(scale, scale_o, rtol=5e-3, atol=1e-4)
    # splicing log-ratios (polee_splicing.py:14-113): feature = the gene's first transcripts, antifeature = the rest
    fpairs, apairs = [], []
    for gidx in range(F):
        ts = np.flatnonzero(gene_of == gidx)
        if len(ts) >= 2:
            fpairs += [(gidx, int(t_)) for t_ in ts[:len(ts) // 2]]
            apairs += [(gidx, int(t_)) for t_ in ts[len(ts) // 2:]]
    has = np.unique([p_[0] for p_ in fpairs])
    sl, ss = ap.approximate_splicing_likelihood(F, np.array(fpairs), np.array(apairs), D1, D2, z0=z0)
    lr = []
    for d in range(D1 + D2):
        x = O.tf_sampler(z0[d], eff, mu, sigma, alpha, L_, R_, F_).astype(np.float64)
        fa = np.zeros((S, F)); an = np.zeros((S, F))
        for (gq, tq) in fpairs: fa[:, gq] += x[:, tq]
        for (gq, tq) in apairs: an[:, gq] += x[:, tq]
        with np.errstate(divide="ignore", invalid="ignore"):
            lr.append(np.log(fa) - np.log(an))
    lr = np.array(lr)
    sl_o = lr[:D1].mean(axis=0); ss_o = np.sqrt(((sl_o[None] - lr[D1:]) ** 2).mean(axis=0))
    np.testing.assert_allclose(sl[:, has], sl_o[:, has], rtol=2e-4, atol=3e-4)
    np.testing.assert_allclose(ss[:, has], ss_o[:, has], rtol=5e-3, atol=2e-4)
    # device RNG, more draws: consistent with a second, independent estimate
    a1, b1 = ap.approximate_feature_likelihood(F, fi, ti, 300, 300, seed=1)
    a2, b2 = ap.approximate_feature_likelihood(F, fi, ti, 300, 300, seed=2)
    assert (np.abs(a1 - a2) < 6 * np.maximum(b1, b2) / np.sqrt(300) + 1e-3).all()
    assert np.median(np.abs(b1 - b2) / b1) < 0.15


def test_elementwise_reparameterisations_match_oracle(P, ctx):
    """Standalone logit-normal / sinh-arcsinh / Kumaraswamy transforms and their gradients."""
    rng = np.random.default_rng(31)
    k = 5000
    mu = rng.normal(size=k).astype(np.float32)
    sigma = np.exp(rng.normal(-1, 0.5, size=k)).astype(np.float32)
    alpha = rng.normal(0, 0.3, size=k).astype(np.float32)
    z0 = rng.normal(size=k).astype(np.float32)
    yg = rng.normal(size=k).astype(np.float32)
    zs, skew = P.sinh_asinh_transform(alpha, z0, True, ctx=ctx)
    zo, skew_o = O.sinh_asinh_transform(alpha, z0, True)
    np.testing.assert_allclose(zs, zo, rtol=2e-6, atol=1e-7)
    assert abs(skew - skew_o) <= 1e-4 * max(1, abs(skew_o))  # the reference accumulates this ladj in f32
    ys, ln = P.logit_normal_transform(mu, sigma, zo, True, ctx=ctx)
    yo, ln_o = O.logit_normal_transform(mu, sigma, zo, True)
    np.testing.assert_allclose(ys, yo, rtol=5e-7)
    assert abs(ln - ln_o) <= 1e-4 * max(1, abs(ln_o))
    zg, mg, sg = P.logit_normal_transform_gradients(zo, yo, mu, sigma, yg, ctx=ctx)
    zgo, mgo, sgo = O.logit_normal_transform_gradients(zo, yo, mu, sigma, yg)
    for a, b in ((zg, zgo), (mg, mgo), (sg, sgo)):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
    ag = P.sinh_asinh_transform_gradients(z0, alpha, zgo, ctx=ctx)
    np.testing.assert_allclose(ag, O.sinh_asinh_transform_gradients(z0, alpha, zgo), rtol=5e-6, atol=1e-6)
    a = rng.uniform(0.5, 3, k).astype(np.float32)
    b = rng.uniform(0.5, 3, k).astype(np.float32)
    z = rng.uniform(0.02, 0.98, k).astype(np.float32)
    yk, lk = P.kumaraswamy_transform(a, b, z, True, ctx=ctx)
    yko, lko = O.kumaraswamy_transform(a, b, z, True)
    np.testing.assert_allclose(yk, yko, rtol=1e-12)
    assert abs(lk - lko) <= 1e-10 * max(1, abs(lko))
    ka, kb = P.kumaraswamy_transform_gradients(z, a, b, yg, ctx=ctx)
    kao, kbo = O.kumaraswamy_transform_gradients(z, a, b, yg)
    np.testing.assert_allclose(ka, kao, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(kb, kbo, rtol=1e-6, atol=1e-6)


def test_optimize_ptt_matches_oracle(P, ctx, lm_fixture):
    """approximate_likelihood(::OptimizePTTApprox) on the :sequential (spine) tree, 40 ADAM steps."""
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    par, js = P.list_nodes(f["n"])
    po, jo = O.list_nodes(f["n"])
    assert (par == po).all() and (js == jo).all()
    t = P.PolyaTreeTransform(par, js, ctx=ctx)
    to = O.PTT(par, js)
    got = P.optimize_likelihood(s, t, num_steps=40)
    xo, zo = O.optimize_ptt(so, to, f["effective_lengths"], num_steps=40)
    # ADAM's clamped first steps make near-zero gradients sign-sensitive: compare what is well conditioned
    # (the oracle, like the reference, carries f32 intermediates down a 312-deep spine: ptt.jl:62)
    ok = np.abs(got["z"] - zo) <= 2e-3 * (1 + np.abs(zo))
    assert ok.mean() > 0.93, ok.mean()
    lp_g, _ = so.log_likelihood(got["x"])
    lp_o, _ = so.log_likelihood(xo)
    assert abs(lp_g - lp_o) <= 1e-4 * abs(lp_o)
    full = P.optimize_likelihood(s, t)  # 500 steps
    lp_full, _ = so.log_likelihood(full["x"])
    assert -327600 < lp_full < -326990, lp_full


def test_row_sharded_fit_with_one_rank_equals_the_plain_fit(P, ctx, lm_fixture, prep_fixture):
    """polee_comm (RCCL) with a single rank: the all-reduce is the identity, so the fit must not change; the
    host-buffer all-reduce returns its input."""
    f = lm_fixture
    s = _gpu_sample(P, ctx, f)
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    comm = P.Comm(ctx, 1, 0)
    v = np.arange(10, dtype=np.float32)
    np.testing.assert_array_equal(comm.allreduce_sum(v), v)
    out = []
    for c in (None, comm):
        fit = P.LikelihoodApproximationFit(s, t, num_steps=8, num_mc_samples=3, seed=5, comm=c, gradonly=False)
        fit.run(8)
        fit.sync()
        out.append((fit.params(), fit.trace()))
    # (two runs of the same fit differ in the last bits: the gradient is accumulated with f32 atomics)
    for a, b in zip(out[0][0], out[1][0]):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out[0][1][1], out[1][1][1], rtol=1e-6)  # expected log-likelihood trace


def test_reduce_scatter_all_gather_exchange_with_one_rank():
    """POLEE_COMM_ALGO=rs_ag (comm.cpp): the exchange as ncclReduceScatter + ncclAllGather, in place.  With one rank both are
    the identity; what runs here is the call path -- RCCL symbols bound, in-place offsets, a count the ranks divide and one
    they do not (1 divides everything: the second call is the same path) -- on the only transport a one-GPU box has.  The
    library reads the switch once per process: a child process."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys; sys.path.insert(0, %r)
        import numpy as np
        import polee_amd as P
        ctx = P.Context(0)
        comm = P.Comm(ctx, 1, 0)
        assert comm.info()["algo"] == "rs_ag" and comm.info()["transport"] == "rccl", comm.info()
        for n in (10, 1200000, 7):
            v = np.random.default_rng(n).normal(size=n).astype(np.float32)
            np.testing.assert_array_equal(comm.allreduce_sum(v), v)
        print("ok")
    """) % root
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, POLEE_COMM_ALGO="rs_ag"), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]


def test_handles_release_device_memory(P, lm_fixture, prep_fixture):
    """create / destroy cycles of every handle type (in arbitrary finaliser order) give the device memory back."""
    import gc
    f = lm_fixture
    probe = P.Context(0)

    def free_bytes():
        gc.collect()
        probe.synchronize()
        return probe.mem_info()[0]

    def cycle():
        ctx = P.Context(0)
        s = P.RNASeqSample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"], f["effective_lengths"], ctx=ctx)
        t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
        fit = P.LikelihoodApproximationFit(s, t, num_steps=3, num_mc_samples=6)
        fit.run(3)
        fit.sync()
        comm = P.Comm(ctx, 1, 0)
        fit2 = P.LikelihoodApproximationFit(s, t, num_steps=2, num_mc_samples=2, comm=comm)
        fit2.run(2)
        fit2.sync()
        # drop in an order that differs from creation order: parents before children
        del ctx, s, t
        del comm
        del fit2, fit

    cycle()
    cycle()  # the first cycles pay one-off allocations (code objects, RCCL and allocator pools: ~56 MB, then flat)
    base = free_bytes()
    for _ in range(5):
        cycle()
    assert base - free_bytes() < 8 << 20, (base, free_bytes())


def test_prep_from_likelihood_matrix_file_to_prepared_sample_file(P, ctx, lm_fixture, prep_fixture, tmp_path):
    """`python -m polee_amd.prep`: likelihood-matrix HDF5 -> (hclust, fit on the GPU) -> prepared-sample HDF5 that the
    model-entry loader reads back; the written approximation explains the data like the reference's own file."""
    from polee_amd import h5io, prep
    f = lm_fixture
    lm_file, out_file = str(tmp_path / "lm.h5"), str(tmp_path / "prep.h5")
    h5io.write_likelihood_matrix(lm_file, f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"], f["effective_lengths"])
    assert prep.main([lm_file, "-o", out_file, "--seed", "7"]) == 0
    got = h5io.read_prepared_sample(out_file)  # (checks the format version)
    assert got["n"] == f["n"] and got["m"] == f["m"] and len(got["node_js"]) == 2 * f["n"] - 1
    np.testing.assert_array_equal(got["effective_lengths"], f["effective_lengths"])
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    l = f["effective_lengths"]
    lp_ref, _ = _expected_loglik(so, O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"]),
                                 prep_fixture["mu"], prep_fixture["omega"], prep_fixture["alpha"], l, 60, 1000)
    lp_fit, _ = _expected_loglik(so, O.PTT(got["node_parent_idxs"], got["node_js"]), got["mu"], got["omega"],
                                 got["alpha"], l, 60, 9000)
    assert lp_fit.mean() > lp_ref.mean() - (6 * np.hypot(lp_ref.std(), lp_fit.std()) / np.sqrt(60) + 10)
    # with the reference's tree given as a --ptt-tree file
    tree_file = str(tmp_path / "tree.h5")
    with h5io.File(tree_file, "w") as h:
        h.write("node_parent_idxs", np.ascontiguousarray(prep_fixture["node_parent_idxs"], np.int32))
        h.write("node_js", np.ascontiguousarray(prep_fixture["node_js"], np.int32))
    assert prep.main([lm_file, "-o", out_file, "--ptt-tree", tree_file]) == 0
    got2 = h5io.read_prepared_sample(out_file)
    np.testing.assert_array_equal(got2["node_js"], prep_fixture["node_js"])


def test_degenerate_sizes(P, ctx):
    """Smallest problems through the whole C ABI: two transcripts (one internal node), a sample without fragments,
    a single fragment; nothing may crash and everything stays finite."""
    rng = np.random.default_rng(41)
    # n = 2: the tree is one internal node with two leaves
    parents, js = np.array([0, 1, 1], np.int32), np.array([0, 1, 2], np.int32)
    t = P.PolyaTreeTransform(parents, js, ctx=ctx)
    x, ladj = t.transform(np.array([0.25]), compute_ladj=True)
    np.testing.assert_allclose(np.sort(x), [0.25, 0.75], rtol=1e-6)
    m, n = 7, 2
    import scipy.sparse as sp
    D = np.where(rng.random((m, n)) < 0.7, rng.uniform(1e-5, 1e-3, (m, n)), 0).astype(np.float32)
    D[D.sum(axis=1) == 0, 0] = 1e-4  # every fragment is compatible with something (compact_indexes!, rnaseq_sample.jl:126)
    X = sp.csc_matrix(D); X.sort_indices(); X.eliminate_zeros()
    colptr, rowval, nzval = (X.indptr + 1).astype(np.uint32), (X.indices + 1).astype(np.uint32), X.data.astype(np.float32)
    eff = np.array([500.0, 900.0], np.float32)
    s = P.RNASeqSample(m, n, colptr, rowval, nzval, eff, ctx=ctx)
    so = O.Sample(m, n, colptr, rowval, nzval)
    xv = np.array([0.3, 0.7], np.float32)
    lp, g = s.log_likelihood(xv)
    lpo, go = so.log_likelihood(xv)
    assert abs(lp - lpo) < 1e-6 * abs(lpo)
    np.testing.assert_allclose(g, go, rtol=3e-5)
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=30)
    assert all(np.isfinite(got[k]).all() and got[k].shape == (1,) for k in ("mu", "omega", "alpha"))
    # a sample without any fragment: the likelihood is constant
    s0 = P.RNASeqSample(0, 2, np.array([1, 1, 1], np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32), eff, ctx=ctx)
    lp0, g0 = s0.log_likelihood(xv)
    assert lp0 == 0 and not g0.any()
    got0 = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s0, t, num_steps=5)
    assert np.isfinite(got0["mu"]).all()
    # hclust on two transcripts, one of them without reads
    p2, j2 = P.hclust(3, 2, np.array([1, 3, 3], np.uint32), np.array([1, 2], np.uint32))
    assert sorted(j2.tolist()) == [0, 1, 2] and p2.tolist() == [0, 1, 1]


def test_factored_vi_trajectory_with_supplied_noise(P, ctx, lm_fixture, prep_fixture):
    """The factored variant of the fit (likelihood-approximation.jl:248-392: X with row multiplicities ks, a given
    tree): four iterations with identical noise on both sides."""
    f = lm_fixture
    rng = np.random.default_rng(17)
    ks = rng.integers(1, 6, f["m"]).astype(np.int64)
    s = P.RNASeqSample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"], f["effective_lengths"], ks=ks, ctx=ctx)
    t = P.PolyaTreeTransform(prep_fixture["node_parent_idxs"], prep_fixture["node_js"], ctx=ctx)
    so = O.Sample(f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"])
    to = O.PTT(prep_fixture["node_parent_idxs"], prep_fixture["node_js"])
    steps, K = 4, 6
    z0 = O.randn(steps * K * (f["n"] - 1), 5)
    ref = O.approximate_likelihood(so, to, f["effective_lengths"], num_steps=steps, num_mc=K, z0=z0, gradonly=False, ks=ks)
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=steps, num_mc_samples=K, z0=z0,
                                   gradonly=False)
    for key in ("mu", "omega", "alpha"):
        ok = np.abs(got[key] - ref[key]) <= 2e-4 * (1 + np.abs(ref[key]))
        assert ok.mean() >= 0.99, (key, ok.mean())
    np.testing.assert_allclose(got["lp_mean"], ref["lp_mean"], rtol=1e-5)
    # the multiplicities matter: the unweighted fit sees a different likelihood
    ref1 = O.approximate_likelihood(so, to, f["effective_lengths"], num_steps=1, num_mc=K, z0=z0[:K * (f["n"] - 1)], gradonly=False)
    assert abs(ref1["lp_mean"][0] - ref["lp_mean"][0]) > 0.1 * abs(ref1["lp_mean"][0])


def test_fast_log_is_a_double_precision_log(P, ctx):
    """The tree kernels take log y, log(1 - y), log u per node in double; csrc/scan.hpp's fast_log replaces libm's
    (a third of the f64 instructions).  It must stay a double-precision log: <= 4 ulp over the whole range."""
    import ctypes as C
    from polee_amd import _lib as L
    rng = np.random.default_rng(40)
    x = np.concatenate([rng.uniform(0, 1, 200000), 10.0 ** rng.uniform(-300, 300, 200000),
                        1 - 10.0 ** rng.uniform(-16, -1, 50000), 1 + 10.0 ** rng.uniform(-16, -1, 50000),
                        [1.0, 0.5, 2.0, 0.7071067811865476, 1e-10, 1 - 1e-10, 5e-324, 1.7976931348623157e308]])
    out = np.empty_like(x)
    L.check(L.lib().polee_debug_fast_log(ctx._h, x.ctypes.data_as(L.f64p), C.c_int64(x.size), out.ctypes.data_as(L.f64p)),
            ctx._h)
    ref = np.log(x)
    ulp = np.abs(out - ref) / np.maximum(np.spacing(np.abs(ref)), 5e-324)
    assert out[x == 1.0].max() == 0.0 and np.all(np.isfinite(out))
    assert ulp.max() <= 4, (ulp.max(), x[ulp.argmax()])
    edge = np.array([0.0, -1.0, np.inf, np.nan])
    eo = np.empty_like(edge)
    L.check(L.lib().polee_debug_fast_log(ctx._h, edge.ctypes.data_as(L.f64p), C.c_int64(4), eo.ctypes.data_as(L.f64p)), ctx._h)
    assert eo[0] == -np.inf and np.isnan(eo[1]) and eo[2] == np.inf and np.isnan(eo[3])


def test_fast_exp_is_a_double_precision_exp(P, ctx):
    """The VI loop's forward kernel exponentiates a path sum of edge logs per leaf and draw (csrc/scan.hpp's fast_exp, under half
    of libm's instructions): <= 2 ulp wherever the result is a normal number, 0 below the underflow threshold, NaN kept."""
    import ctypes as C
    from polee_amd import _lib as L
    rng = np.random.default_rng(41)
    x = np.concatenate([-10.0 ** rng.uniform(-17, 2.85, 300000), rng.uniform(-1, 1, 50000), rng.uniform(-40, 0, 100000),
                        [0.0, -0.0, -1e-300, -0.34657359027997264, 0.34657359027997264, -708.0, 5.0, 100.0]])
    out = np.empty_like(x)
    L.check(L.lib().polee_debug_fast_exp(ctx._h, x.ctypes.data_as(L.f64p), C.c_int64(x.size), out.ctypes.data_as(L.f64p)),
            ctx._h)
    ref = np.exp(x)
    normal = ref > 2.3e-308
    ulp = np.abs(out - ref)[normal] / np.spacing(ref[normal])
    assert out[x == 0.0].min() == 1.0 and ulp.max() <= 2, (ulp.max(), x[normal][ulp.argmax()])
    assert np.all(np.abs(out - ref)[~normal] <= 1e-307)
    edge = np.array([-746.0, -1e6, -np.inf, np.nan])
    eo = np.empty_like(edge)
    L.check(L.lib().polee_debug_fast_exp(ctx._h, edge.ctypes.data_as(L.f64p), C.c_int64(4), eo.ctypes.data_as(L.f64p)), ctx._h)
    assert eo[0] == 0.0 and eo[1] == 0.0 and eo[2] == 0.0 and np.isnan(eo[3])


def test_cohort_pipeline_matches_sequential_fits(P, lm_fixture):
    """approximate_likelihood_cohort: several samples in flight on one GPU (a worker thread and a HIP stream each, host
    stages of one under the device stage of another) give what one-at-a-time calls give: the same tree node for node,
    and the same fitted parameters up to the float32 summation order of the likelihood's atomics."""
    f = lm_fixture
    rng = np.random.default_rng(11)
    samples = []
    for i in range(5):  # the fixture's matrix with perturbed probabilities: different samples, same support
        nz = (f["nzval"] * rng.uniform(0.5, 1.5, size=f["nzval"].shape)).astype(np.float32)
        samples.append((f["m"], f["n"], f["colptr"], f["rowval"], nz, f["effective_lengths"]))
    approx = P.LogitSkewNormalPTTApprox("cluster")
    kw = dict(num_steps=30, num_mc_samples=6, seed=5)
    seq = []
    for s in samples:
        ctx = P.Context(0)
        seq.append(P.approximate_likelihood(approx, P.RNASeqSample(*s, ctx=ctx), **kw))
    calls = []
    lazy = [(lambda s=s: (calls.append(1), s)[1]) for s in samples]  # sources may be callables (loaded inside the worker)
    par = P.approximate_likelihood_cohort(approx, lazy, workers=3, **kw)
    assert len(par) == len(samples) and len(calls) == len(samples)
    for a, b in zip(seq, par):
        assert (a["node_parent_idxs"] == b["node_parent_idxs"]).all() and (a["node_js"] == b["node_js"]).all()
        for key in ("mu", "omega", "alpha"):
            np.testing.assert_allclose(b[key], a[key], rtol=0, atol=2e-3)
    got = {}
    as_dicts = [dict(zip(("m", "n", "colptr", "rowval", "nzval", "effective_lengths"), s)) for s in samples[:2]]  # (h5io.read_likelihood_matrix)
    assert P.approximate_likelihood_cohort(approx, as_dicts, workers=2, on_result=lambda i, p: got.__setitem__(i, p), **kw) == [None, None]
    assert sorted(got) == [0, 1] and np.isfinite(got[1]["mu"]).all()


def test_cohort_of_worker_processes_matches_sequential_fits(P, lm_fixture, tmp_path):
    """approximate_likelihood_cohort_processes: one worker PROCESS per sample in flight (`spawn`; each its own HIP context
    on the one GPU), samples read inside the workers from their files; the parallel tree variant; the same trees and --
    up to the float32 summation order of the likelihood's atomics -- the same fitted parameters as one-at-a-time calls."""
    import functools
    f = lm_fixture
    rng = np.random.default_rng(12)
    paths, samples = [], []
    for i in range(3):
        nz = (f["nzval"] * rng.uniform(0.5, 1.5, size=f["nzval"].shape)).astype(np.float32)
        samples.append((f["m"], f["n"], f["colptr"], f["rowval"], nz, f["effective_lengths"]))
        paths.append(str(tmp_path / ("sample%d.npz" % i)))
        np.savez(paths[-1], m=np.array([f["m"]]), n=np.array([f["n"]]), colptr=f["colptr"], rowval=f["rowval"], nzval=nz,
                 effective_lengths=f["effective_lengths"])
    approx = P.LogitSkewNormalPTTApprox("cluster_parallel")
    kw = dict(num_steps=30, num_mc_samples=6, seed=5)
    seq = [P.approximate_likelihood(approx, P.RNASeqSample(*s, ctx=P.Context(0)), **kw) for s in samples]
    par = P.approximate_likelihood_cohort_processes(approx, [functools.partial(np.load, p) for p in paths], processes=2,
                                                    host_threads=2, **kw)
    assert len(par) == 3
    for a, b in zip(seq, par):
        assert (a["node_parent_idxs"] == b["node_parent_idxs"]).all() and (a["node_js"] == b["node_js"]).all()
        for key in ("mu", "omega", "alpha"):
            np.testing.assert_allclose(b[key], a[key], rtol=0, atol=2e-3)


def test_sample_and_tree_side_by_side_equals_one_after_the_other(P, ctx, lm_fixture):
    """core.sample_and_tree builds the tree on a helper thread while the device layout is built: the same tree arrays as
    hclust() alone (both modes) and a sample handle that evaluates like one built on its own."""
    f = lm_fixture
    args = (f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"], f["effective_lengths"])
    x = np.random.default_rng(3).dirichlet(np.ones(f["n"])).astype(np.float32)
    ref = P.RNASeqSample(*args, ctx=ctx)
    lp0, g0 = ref.log_likelihood(x)
    for method, par in (("cluster", False), ("cluster_parallel", True)):
        s, t = P.sample_and_tree(P.LogitSkewNormalPTTApprox(method), *args, ctx=ctx)
        pe, je = P.hclust(f["m"], f["n"], f["colptr"], f["rowval"], parallel=par)
        params = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(method), s, t, num_steps=5, seed=1)
        np.testing.assert_array_equal(params["node_parent_idxs"], pe)
        np.testing.assert_array_equal(params["node_js"], je)
        lp, g = s.log_likelihood(x)
        np.testing.assert_allclose(lp, lp0, rtol=1e-6)
        np.testing.assert_allclose(g, g0, rtol=1e-4, atol=1e-3 * np.abs(g0).max())
    s, t = P.sample_and_tree(P.LogitSkewNormalPTTApprox("sequential"), *args, ctx=ctx)
    assert len(P.approximate_likelihood(P.LogitSkewNormalPTTApprox("sequential"), s, t, num_steps=2)["mu"]) == f["n"] - 1
    with pytest.raises(ValueError):
        P.sample_and_tree(P.LogitSkewNormalPTTApprox("random"), *args, ctx=ctx)
