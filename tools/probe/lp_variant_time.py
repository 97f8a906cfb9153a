"""Sparse-pass time of the gradient-only kernel instance against the one that also returns lp (the ELBO trace)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789)
parents, js = synth.make_tree(smp["gene"], 123456789)
ctx = P.Context(0)
s = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
t = P.PolyaTreeTransform(parents, js, ctx=ctx)
for gradonly in (True, False, True, False):
    f = P.LikelihoodApproximationFit(s, t, num_steps=400, num_mc_samples=6, seed=1, profile=True, gradonly=gradonly)
    f.run(100); f.sync(); st0 = f.stats()
    f.run(200); f.sync(); st1 = f.stats()
    l = st1["loglik_kernel_launches"] - st0["loglik_kernel_launches"]
    k = (st1["loglik_kernel_ms_avg"] * st1["loglik_kernel_launches"] - st0["loglik_kernel_ms_avg"] * st0["loglik_kernel_launches"]) / l
    print("gradonly", gradonly, "kernel ms %.4f" % k)
    del f
