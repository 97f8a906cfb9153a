"""Exact (reference-order) tree against the parallel-rounds tree: construction time and the fit each gives.
usage: hclust_quality.py [fixture] [small] [c1] [c2] [tiled]   (ELBO / E[lp]: mean over the last 100 of 500 steps, 3 seeds)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import polee_amd as P
from tools import synth

def load(which):
    if which == "fixture":
        a = np.load(os.path.join(ROOT, "tests", "golden", "mBr_M_6w_1.likelihood-matrix.npz"))
        return int(a["m"][0]), int(a["n"][0]), a["colptr"], a["rowval"], a["nzval"], a["effective_lengths"]
    if which == "tiled":  # the reference fixture x639: 639 disconnected blocks with the real set structure
        smp = synth.tile_fixture(639)
        colptr, rowval, nzval = synth.to_csc(smp)
        return int(smp["m"]), int(smp["n"]), colptr, rowval, nzval, smp["effective_lengths"]
    n, m = {"small": (20000, 3000000), "c1": (2000, 300000), "c2": (200000, 30000000)}[which]
    smp = synth.make_sample(n, m, 8.0, 123456789)
    colptr, rowval, nzval = synth.to_csc(smp)
    return m, n, colptr, rowval, nzval, smp["effective_lengths"]

ctx = P.Context(0)
for which in sys.argv[1:] or ["fixture", "c1", "small"]:
    m, n, colptr, rowval, nzval, eff = load(which)
    s = P.RNASeqSample(m, n, colptr, rowval, nzval, eff, ctx=ctx)
    for par in (False, True):
        t0 = time.time()
        parents, js = P.hclust(m, n, colptr, rowval, parallel=par)
        th = time.time() - t0
        # depth of the tree (the transform's scans do not care, the fit's conditioning might)
        depth = np.zeros(2 * n - 1, np.int32)
        for i in range(1, 2 * n - 1):
            depth[i] = depth[parents[i] - 1] + 1
        tr = P.PolyaTreeTransform(parents, js, ctx=ctx)
        res = []
        for seed in (1, 2, 3):
            fit = P.LikelihoodApproximationFit(s, tr, num_steps=500, num_mc_samples=6, seed=seed, gradonly=False)
            fit.run(500)
            fit.sync()
            e, l = fit.trace()
            res.append((e[-100:].mean(), l[-100:].mean()))
            del fit
        res = np.array(res)
        print("%-8s %-8s hclust %7.3f s  leaf depth mean %.1f max %d | ELBO %s  mean %.2f | E[lp] %s mean %.2f" % (
            which, "parallel" if par else "exact", th, depth[js > 0].mean(), depth.max(),
            " ".join("%.2f" % v for v in res[:, 0]), res[:, 0].mean(),
            " ".join("%.2f" % v for v in res[:, 1]), res[:, 1].mean()), flush=True)
    del s
