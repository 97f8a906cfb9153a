"""Regression steps/s (SURVEY.md 8(d) secondary metric, configs C3 / C4 per-GPU share): S samples, F=2 factors,
n=200k transcripts, synthetic approximation parameters.  usage: regression_bench.py [S] [steps]"""
import os, sys, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n = 200000
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(0)
smp = synth.make_sample(n, 1000000, 8.0, 1)
par, js = synth.make_tree(smp['gene'], 1)
l, r, f = P.make_inverse_ptt_params(par, js)
eff = np.tile(smp['effective_lengths'], (S, 1)).astype(np.float32)
mu = rng.normal(0, 2, (S, n - 1)).astype(np.float32)
sigma = np.exp(rng.normal(-1, 1, (S, n - 1))).astype(np.float32)
alpha = rng.normal(0, .3, (S, n - 1)).astype(np.float32)
ctx = P.Context(0)
vars_ = dict(efflen=eff, la_mu=mu, la_sigma=sigma, la_alpha=alpha, left_index=l[None], right_index=r[None], leaf_index=f[None])
ap = P.RNASeqApproxLikelihood(vars_, ctx=ctx)
x0 = np.log(np.maximum(ap.sample(seed=1), 1e-12)).astype(np.float32)
design = np.zeros((S, 2), np.float32); design[:, 0] = 1; design[S // 2:, 1] = 1
ss = P.estimate_sample_scales(x0)
for point in (False, True):
    reg = P.RNASeqTranscriptLinearRegression(ap, x0, design, ss, True, 1.0, point, ctx=ctx)
    reg.fit(20)
    t0 = time.perf_counter()
    out = reg.fit(steps, return_trace=True)
    dt = time.perf_counter() - t0
    tr = out[-1]
    print("S=%d F=2 n=%d point_estimates=%s: %.3f ms/step = %.0f steps/s; loss %.4g -> %.4g finite=%s; params %.1f M" % (
        S, n, point, dt / steps * 1e3, steps / dt, tr[0], tr[-1], bool(np.all(np.isfinite(tr))), reg.num_params / 1e6))
