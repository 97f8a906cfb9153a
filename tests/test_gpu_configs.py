"""-m gpu: the HIP path against the CPU oracle AT THE SIZES OF BASELINE.json's configs ("what's weak" 1-3).

C2 (n = 200 000 x m = 30 M, 240 M nnz): log-likelihood and the full per-transcript gradient of K = 6 draws against
the oracle (which accumulates frag_probs and x_grad in float64 like sparse.jl:13-17,32-36 with Vector{Float64}),
and three whole VI iterations with supplied noise against oracle's approximate_likelihood.
C1 (n = 1 000 x m = 100 000, 2.2 nnz / fragment): the WHOLE 500 x 6 fit replayed with the same noise on both sides.
C3 (S = 6, F = 2, n = 200 000): the regression model's loss and gradients against the float64 restatement.
C4's per-GPU share (S = 8) runs through the same kernels as C3 with a longer sample loop: value and gradients.
C5 (m = 150 M, 1.2 G nnz) at K = 6: lp and the full gradient against the oracle run on five blocks of rows.

Tolerances (north_star: 1e-4 relative on the log-likelihood): lp 1e-6 relative; gradient 1e-4 relative + 1e-6 of the
largest entry (the device accumulates the gradient in f32, the oracle in f64: the worst transcript and its column
length are printed)."""
import time

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

N, M, NNZ_PER_FRAG = 200_000, 30_000_000, 8.0


@pytest.fixture(scope="module")
def P():
    import polee_amd
    return polee_amd


@pytest.fixture(scope="module")
def c2(P):
    from tools import synth
    smp = synth.make_sample(N, M, NNZ_PER_FRAG, seed=123456789)
    ctx = P.Context(0)
    s = P.RNASeqSample(M, N, None, None, None, smp["effective_lengths"], ctx=ctx,
                       xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    colptr, rowval, nzval = synth.to_csc(smp)
    so = O.Sample(M, N, colptr, rowval, nzval)
    collen = np.diff(colptr.astype(np.int64))
    del rowval, nzval
    parents, js = synth.make_tree(smp["gene"], seed=3, kind="hclust")
    return ctx, smp, s, so, collen, parents, js


def test_c2_loglik_and_gradient_match_oracle(P, c2):
    ctx, smp, s, so, collen, parents, js = c2
    O.set_num_threads(O.physical_cores())
    rng = np.random.default_rng(0)
    K = 6
    x = rng.gamma(0.3, size=(K, N)).astype(np.float32) + np.float32(1e-7)
    x /= x.sum(axis=1, keepdims=True)
    x = np.clip(x, np.float32(1e-10), 1)
    lp, g = s.log_likelihood(x)
    worst = (0.0, -1, -1)
    for k in range(K):
        lpo, go = so.log_likelihood(x[k])
        assert abs(lp[k] - lpo) <= 1e-6 * abs(lpo), (k, lp[k], lpo)
        scale = np.abs(go).max()
        err = np.abs(g[k] - go) / (np.abs(go) + 1e-2 * scale)
        j = int(np.argmax(err))
        if err[j] > worst[0]:
            worst = (float(err[j]), k, j)
        np.testing.assert_allclose(g[k], go, rtol=1e-4, atol=1e-6 * scale)
    # the transcripts with the longest columns (f32 accumulation over the most fragments)
    longest = np.argsort(collen)[-5:]
    rel_long = np.abs(g[0][longest] - so.log_likelihood(x[0])[1][longest]) / np.abs(g[0][longest])
    print("C2 gradient vs f64 oracle: worst weighted error %.3g (draw %d, transcript %d, column of %d fragments); "
          "5 longest columns (%s fragments): rel err %s" % (worst[0], worst[1], worst[2], collen[worst[2]],
                                                            collen[longest].tolist(), np.array2string(rel_long, precision=2)))
    assert rel_long.max() < 1e-4


@pytest.mark.parametrize("case", ["literal", "dropout 0.3"])
def test_c2_size_set_diverse_inputs_match_oracle(P, case):
    """the C2-size oracle comparison on the inputs the generator does not flatter -- every fragment its
    own random subset of its gene's isoforms (SURVEY 8(d) read literally) and per-entry dropout 0.3 -- which route a large
    share of X through the dense-union, masked and wide streams.  Same tolerances as the C2 test above."""
    from tools import synth
    kw = dict(literal=True) if case == "literal" else dict(dropout=0.3)
    smp = synth.make_sample(N, M, NNZ_PER_FRAG, seed=123456789, **kw)
    ctx = P.Context(0)
    s = P.RNASeqSample(M, N, None, None, None, smp["effective_lengths"], ctx=ctx,
                       xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    info = s.info
    assert sum(info["stream_nnz"]) == smp["nnz"] and sum(info["stream_rows"]) == M and info["stream_tiles"][5] == 0
    assert info["stream_nnz"][1] + info["stream_nnz"][2] + info["stream_nnz"][3] > 0.25 * smp["nnz"]  # not the A1 stream's show
    colptr, rowval, nzval = synth.to_csc(smp)
    so = O.Sample(M, N, colptr, rowval, nzval)
    del colptr, rowval, nzval
    O.set_num_threads(O.physical_cores())
    rng = np.random.default_rng(1)
    K = 6
    x = rng.gamma(0.3, size=(K, N)).astype(np.float32) + np.float32(1e-7)
    x /= x.sum(axis=1, keepdims=True)
    x = np.clip(x, np.float32(1e-10), 1)
    lp, g = s.log_likelihood(x)
    worst = 0.0
    for k in range(K):
        lpo, go = so.log_likelihood(x[k])
        assert abs(lp[k] - lpo) <= 1e-6 * abs(lpo), (case, k, lp[k], lpo)
        scale = np.abs(go).max()
        worst = max(worst, float((np.abs(g[k] - go) / (np.abs(go) + 1e-2 * scale)).max()))
        np.testing.assert_allclose(g[k], go, rtol=1e-4, atol=1e-6 * scale)
    print("C2-size %s: shares of nnz %s, worst weighted gradient error %.3g" % (
        case, [round(v / smp["nnz"], 3) for v in info["stream_nnz"]], worst))


def test_c2_three_vi_iterations_match_oracle(P, c2):
    """Three whole iterations (sampling, tree transform, likelihood over the 240 M non-zeros, backward, ADAM) with the
    same z0 on both sides (likelihood-approximation.jl:496-572)."""
    ctx, smp, s, so, collen, parents, js = c2
    O.set_num_threads(O.physical_cores())
    t = P.PolyaTreeTransform(parents, js, ctx=ctx)
    to = O.PTT(parents, js)
    steps, K = 3, 6
    z0 = O.randn(steps * K * (N - 1), 5)
    ref = O.approximate_likelihood(so, to, smp["effective_lengths"], num_steps=steps, num_mc=K, z0=z0, gradonly=False)
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=steps, num_mc_samples=K, z0=z0,
                                   gradonly=False)
    np.testing.assert_allclose(got["lp_mean"], ref["lp_mean"], rtol=1e-6)
    np.testing.assert_allclose(got["elbo"], ref["elbo"], rtol=1e-6)
    # ADAM's first steps move a parameter by +-max_step whatever its gradient's size: nodes whose gradient is ~0 are
    # sign-sensitive to rounding (the reference keeps f32 intermediates in the tree backward pass, ptt.jl:62)
    for key in ("mu", "omega", "alpha"):
        ok = np.abs(got[key] - ref[key]) <= 2e-4 * (1 + np.abs(ref[key]))
        print("C2 3-step", key, "agree on", ok.mean())
        assert ok.mean() >= 0.99, (key, ok.mean())


@pytest.mark.parametrize("literal", [False, True], ids=["patterns", "literal"])
def test_c5_loglik_and_gradient_match_oracle_at_k6(P, literal):
    """(literal: every fragment its own random subset of its gene's isoforms, SURVEY 8(d) read literally -- the generator of
    bench.py's headline; VERDICT r4 item 6a: the C5 bench line of that input now has its oracle twin.)
    BASELINE configs[4] (n = 200 000 x m = 150 M fragments, 1.2 G non-zeros) at the production K = 6 against the
    ORACLE: lp and x_grad are sums over fragments, so the oracle runs on five blocks of 30 M rows and the blocks are
    added (the arithmetic of the row-sharded fit, tests/test_multiproc.py); the device makes one pass over the whole
    matrix.  lp 1e-6, gradient 1e-4 (+ 1e-6 of the largest entry)."""
    from tools import synth
    from polee_amd.cohort import take_rows
    m5, K, nblocks = 150_000_000, 6, 5
    smp = synth.make_sample(N, m5, NNZ_PER_FRAG, seed=987654321, literal=literal)
    ctx = P.Context(0)
    s = P.RNASeqSample(m5, N, None, None, None, smp["effective_lengths"], ctx=ctx,
                       xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    rng = np.random.default_rng(5)
    x = rng.gamma(0.3, size=(K, N)).astype(np.float32) + np.float32(1e-7)
    x /= x.sum(axis=1, keepdims=True)
    x = np.clip(x, np.float32(1e-10), 1)
    lp, g = s.log_likelihood(x)
    del s
    O.set_num_threads(O.physical_cores())
    lpo, go = np.zeros(K), np.zeros((K, N))
    step = m5 // nblocks
    for b in range(nblocks):
        r0, r1 = b * step, m5 if b == nblocks - 1 else (b + 1) * step
        tp, tr, tv = take_rows(smp["tcolptr"], smp["trowval"], smp["tnzval"], r0, r1)
        blk = dict(m=r1 - r0, n=N, nnz=int(tr.size), tcolptr=tp, trowval=tr, tnzval=tv)
        colptr, rowval, nzval = synth.to_csc(blk)
        so = O.Sample(r1 - r0, N, colptr, rowval, nzval)
        for k in range(K):
            a, b_ = so.log_likelihood(x[k])
            lpo[k] += a
            go[k] += b_
        del so, colptr, rowval, nzval
    worst = 0.0
    for k in range(K):
        assert abs(lp[k] - lpo[k]) <= 1e-6 * abs(lpo[k]), (k, lp[k], lpo[k])
        scale = np.abs(go[k]).max()
        worst = max(worst, float((np.abs(g[k] - go[k]) / (np.abs(go[k]) + 1e-2 * scale)).max()))
        np.testing.assert_allclose(g[k], go[k], rtol=1e-4, atol=1e-6 * scale)
    print("C5 K=6 (%s) vs the f64 oracle (five 30 M-row blocks): worst weighted gradient error %.3g" % ("literal" if literal else "patterns", worst))


def test_c1_whole_fit_replayed_against_oracle(P):
    """BASELINE configs[0]: 1k transcripts x 100k fragments, the whole 500-iteration x 6-draw fit with supplied noise on
    both sides.  The per-iteration E[lp] trace must agree to 1e-5 all the way; the fitted parameters agree on nearly
    every node (ADAM turns a gradient of ~0 into a +-step, so a few nodes' paths split by rounding and stay split)."""
    from tools import synth
    n, m = 1000, 100_000
    smp = synth.make_sample(n, m, 2.2, seed=123456789)
    parents, js = synth.make_tree(smp["gene"], seed=3, kind="hclust")
    ctx = P.Context(0)
    s = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx,
                       xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    t = P.PolyaTreeTransform(parents, js, ctx=ctx)
    colptr, rowval, nzval = synth.to_csc(smp)
    so, to = O.Sample(m, n, colptr, rowval, nzval), O.PTT(parents, js)
    steps, K = 500, 6
    z0 = O.randn(steps * K * (n - 1), 17)
    ref = O.approximate_likelihood(so, to, smp["effective_lengths"], num_steps=steps, num_mc=K, z0=z0, gradonly=False)
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, num_steps=steps, num_mc_samples=K, z0=z0,
                                   gradonly=False)
    rel = np.abs(got["lp_mean"] - ref["lp_mean"]) / np.abs(ref["lp_mean"])
    print("C1 500-step replay: max rel dev of the E[lp] trace %.3g (final %.3g)" % (rel.max(), rel[-1]))
    assert rel.max() < 1e-4 and rel[-1] < 1e-5
    for key in ("mu", "omega", "alpha"):
        d = np.abs(got[key] - ref[key])
        ok = d <= 1e-3 * (1 + np.abs(ref[key]))
        print("C1 500-step replay:", key, "within 1e-3 on %.4f of the nodes, median |d| %.3g" % (ok.mean(), np.median(d)))
        assert ok.mean() >= 0.97, (key, ok.mean())
    # and the two fitted approximations are the same distribution: posterior means of the sampler
    sg, sr = np.exp(got["omega"]), np.exp(ref["omega"])
    pm_g, pm_r = np.zeros(n), np.zeros(n)
    for d in range(50):
        zz = O.randn(n - 1, 900 + d)
        pm_g += O.sampler_draw(to, got["mu"], sg, got["alpha"], zz)
        pm_r += O.sampler_draw(to, ref["mu"], sr, ref["alpha"], zz)
    big = pm_r / 50 > 1e-5
    np.testing.assert_allclose(pm_g[big], pm_r[big], rtol=2e-3)


def _c3_problem(P, S, F, n, rng):
    from tools import synth
    smp = synth.make_sample(n, 200_000, 8.0, 1)
    parents, js = synth.make_tree(smp["gene"], 1)
    li, ri, fi = P.make_inverse_ptt_params(parents, js)
    vars_ = dict(efflen=np.tile(smp["effective_lengths"], (S, 1)).astype(np.float32),
                 la_mu=rng.normal(0, 2, (S, n - 1)).astype(np.float32),
                 la_sigma=np.exp(rng.normal(-1, 1, (S, n - 1))).astype(np.float32),
                 la_alpha=rng.normal(0, .3, (S, n - 1)).astype(np.float32),
                 left_index=np.tile(li, (S, 1)), right_index=np.tile(ri, (S, 1)), leaf_index=np.tile(fi, (S, 1)))
    design = np.zeros((S, F), np.float32)
    design[:, 0] = 1
    design[S // 2:, 1] = 1
    x_init = (rng.normal(-np.log(n), 1.5, size=(1, n)) + rng.normal(0, 0.4, size=(S, n))).astype(np.float32)
    return vars_, design, x_init


@pytest.mark.parametrize("S,full", [(6, True), (8, True)])
def test_c3_regression_eval_matches_restatement_at_200k(P, S, full):
    """BASELINE configs[2] (S = 6 samples, F = 2 factors, n = 200 000; full = loss + gradients) and the per-GPU share of
    configs[3] (S = 8, also loss + gradients): `polee_regression_eval` against oracle/regression_ref.py + the C oracle's
    approximate likelihood.  Gradient: central differences of the float64 restatement (plus the oracle's analytic
    likelihood gradient) on every scalar parameter, a random subset of the hinge coefficients and of every per-column
    block."""
    from oracle import regression_ref as RR
    rng = np.random.default_rng(41)
    F, n, deg, pen = 2, N, 15, 1.0
    O.set_num_threads(O.physical_cores())
    ctx = P.Context(0)
    vars_, design, x_init = _c3_problem(P, S, F, n, rng)
    ss = P.estimate_sample_scales(x_init)
    reg = P.RNASeqTranscriptLinearRegression(vars_, x_init, design, ss, True, pen, False, ctx=ctx)
    mean = x_init.astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float64)
    W = RR.kernel_regression_weights(1.0, mean, RR.choose_knots(mean.min(), mean.max(), deg))
    p0 = RR.flatten(RR.initial_params(x_init, F, deg), RR.PARAMS)
    theta = (p0 + rng.normal(0, 0.2, size=p0.size)).astype(np.float32)
    reg.set_flat_params(theta)
    eps = rng.normal(size=reg.num_noise).astype(np.float32)
    loss, g = reg.loss_and_gradients(noise=eps)
    a = (vars_["efflen"], vars_["la_mu"], vars_["la_sigma"], vars_["la_alpha"], vars_["left_index"],
         vars_["right_index"], vars_["leaf_index"])
    lik = lambda x: O.approx_log_prob(x.astype(np.float32), *a).astype(np.float64)
    common = dict(design=design.astype(np.float64), W=W, sample_scales=ss, x_bias_loc0=np.log(1.0 / n),
                  x_bias_scale0=12.0, use_distortion=True, scale_penalty=pen, use_point_estimates=False)
    e = RR.unflatten(eps.astype(np.float64), RR.NOISE, S, F, n, deg)
    t64 = theta.astype(np.float64)

    def L_rest(vec, with_lik):
        return RR.regression_loss(RR.unflatten(vec, RR.PARAMS, S, F, n, deg), e, lik=lik if with_lik else None, **common)

    t0 = time.time()
    loss_o, z = L_rest(t64, True)
    print("C3-size restatement: loss %.6g vs device %.6g (%.1f s)" % (loss_o, loss, time.time() - t0))
    assert np.isfinite(loss) and abs(loss - loss_o) <= 1e-4 * abs(loss_o), (loss, loss_o)
    if not full:
        return
    glik = O.approx_log_prob(z["x"].astype(np.float32), *a, want_grad=True)[1].astype(np.float64)
    table = RR.unflatten(np.arange(t64.size), RR.PARAMS, S, F, n, deg)
    check_idx = set(range(4))
    check_idx.update(rng.choice(np.arange(4, 4 + F * deg + 2 * deg), size=6, replace=False).tolist())
    for name, _ in RR.PARAMS[7:]:
        ids = table[name].reshape(-1).astype(int)
        check_idx.update(rng.choice(ids, size=2, replace=False).tolist())
    o_loc, o_s = int(table["qx_loc"].reshape(-1)[0]), int(table["qx_softplus_scale"].reshape(-1)[0])
    gscale = np.abs(g).max()
    worst = 0.0
    for i in sorted(check_idx):
        h = 1e-4 * max(1.0, abs(t64[i]))
        tp, tm = t64.copy(), t64.copy()
        tp[i] += h
        tm[i] -= h
        fd = (L_rest(tp, False)[0] - L_rest(tm, False)[0]) / (2 * h)
        if o_loc <= i < o_loc + S * n:
            fd -= glik.reshape(-1)[i - o_loc]
        elif o_s <= i < o_s + S * n:
            k = i - o_s
            fd -= glik.reshape(-1)[k] * e["x"].reshape(-1)[k] / (1.0 + np.exp(-t64[i]))
        err = abs(g[i] - fd) / (abs(fd) + 2e-3 * gscale)
        worst = max(worst, err)
        assert err < 1e-2, (i, g[i], fd)
    print("C3-size gradient: %d parameters checked, worst weighted error %.3g" % (len(check_idx), worst))
