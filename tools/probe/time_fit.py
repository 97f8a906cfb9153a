"""ms per VI iteration of the C2 fit by HIP events, without checking the results (for timing experiments with the
POLEE_DBG_ABLATE switches, whose results are invalid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import polee_amd as P
from tools import synth
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789)
parents, js = synth.make_tree(smp["gene"], 123456789)
ctx = P.Context(0)
sample = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
tree = P.PolyaTreeTransform(parents, js, ctx=ctx)
fit = P.LikelihoodApproximationFit(sample, tree, num_steps=1000, num_mc_samples=6, seed=1, profile=True)
fit.run(300)
ctx.synchronize()
st0 = fit.stats()
ctx.timer_start()
fit.run(100)
ms = ctx.timer_stop()
st1 = fit.stats()
l = st1["loglik_kernel_launches"] - st0["loglik_kernel_launches"]
k = (st1["loglik_kernel_ms_avg"] * st1["loglik_kernel_launches"] - st0["loglik_kernel_ms_avg"] * st0["loglik_kernel_launches"]) / max(l, 1)
print("ABLATE=%s: %.4f ms per iteration, sparse kernel %.4f ms" % (os.environ.get("POLEE_DBG_ABLATE", "0"), ms / 100, k))
