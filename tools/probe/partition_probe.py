"""Does a fit run on a CU-masked stream (polee_ctx_create_partition)?  usage: partition_probe.py <parts> <which: one|all> [c2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
parts = int(sys.argv[1]); which = sys.argv[2]; big = len(sys.argv) > 3
smp = synth.make_sample(200000, 30000000, 8.0, 1) if big else synth.tile_fixture(5)
parents, js = synth.make_tree(smp["gene"], 1)
fits = []
for p in ([0] if which == "one" else range(parts)):
    ctx = P.Context(0, partition=(p, parts))
    s = P.RNASeqSample(smp["m"], smp["n"], None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    t = P.PolyaTreeTransform(parents, js, ctx=ctx)
    fits.append((ctx, s, t, P.LikelihoodApproximationFit(s, t, num_steps=100, num_mc_samples=6, seed=1, profile=True)))
    print("partition", p, "of", parts, "created", flush=True)
t0 = time.time()
for f in fits:
    f[3].run(100)
for f in fits:
    f[3].sync()
print("ok: %d fit(s) x 100 steps in %.3f s; kernel %.4f ms" % (len(fits), time.time() - t0, fits[0][3].stats()["loglik_kernel_ms_avg"]), flush=True)
