"""X construction on the GPU at scale (polee_xbuild_run): n transcripts, m synthetic alignment pairs; prints the kernel
times and the rate.  usage (GPU box): python3 tools/probe/xbuild_bench.py [n] [m]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import polee_amd as P
from polee_amd import xbuild as XB
from tools import synth_aln
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 30000000
t0 = time.time()
d = synth_aln.make(n, m, num_seq=24, seed=7)
pmf, cdf, med = synth_aln.fraglen_model()
print("generated %d transcripts, %d alignment pairs in %.1f s" % (n, m, time.time() - t0))
ctx = P.Context(0)
for rep in range(2):
    t0 = time.time()
    g = XB.build_likelihood_matrix(d["transcripts"], d["fragments"], pmf, cdf, med, 0.9, False, ctx=ctx)
    wall = time.time() - t0
k = g["kernel_ms"]
tt = d["true_transcript"]
ptr = g["tcolptr"].astype(np.int64) - 1
hit = sum((tt[i] + 1) in g["trowval"][ptr[r]:ptr[r + 1]] for r, i in list(enumerate(g["row_fragment"]))[::max(1, g["m"] // 20000)])
print("rows %d (%.1f %% of the pairs), non-zeros %d (%.2f per row); kernels: effective lengths %.2f ms, count %.2f ms, fill %.2f ms "
      "= %.1f M pairs/s; whole call incl. upload / download %.2f s; source transcript present in %d of %d sampled rows"
      % (g["m"], 100.0 * g["m"] / m, g["nnz"], g["nnz"] / g["m"], k["efflen"], k["count"], k["fill"],
         m / (k["count"] + k["fill"]) / 1e3, wall, hit, len(range(0, g["m"], max(1, g["m"] // 20000)))))

# ---- the whole device pipeline: alignment pairs -> X -> tree + layout -> 500-step fit, X never on the host
if os.environ.get("POLEE_XB_PIPELINE", "1") != "0":
    for rep in range(3):
        t = [time.time()]
        g = XB.build_likelihood_matrix(d["transcripts"], d["fragments"], pmf, cdf, med, 0.9, False, ctx=ctx, return_sample=True, return_tree=True)
        t.append(time.time())
        tr = P.PolyaTreeTransform(g["node_parent_idxs"], g["node_js"], ctx=ctx)
        fit = P.LikelihoodApproximationFit(g["sample"], tr, num_steps=500, num_mc_samples=6, seed=rep)
        t.append(time.time())
        fit.run(500); fit.sync()
        t.append(time.time())
        mu = fit.params()
        del fit, tr, g
        t.append(time.time())
        dd = np.diff(t)
        print("alignment pairs -> X -> tree + layout (incl. the download of X for the caller) %.3f  handles %.3f  500-step fit %.3f  params+free %.3f | total %.3f s"
              % (tuple(dd) + (dd.sum(),)), flush=True)
