"""One C2 sample, stage by stage (single worker): tree (exact / parallel), sample handle (layout build + upload), tree
handle, fit handle, 500-step fit, parameters.  POLEE_BUILD_TIMING=1 adds the builders' own phase times on stderr."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n, m = 200000, int(os.environ.get("POLEE_PREP_M", "30000000"))  # (POLEE_PREP_M=150000000: BASELINE's C5)
smp = synth.make_sample(n, m, 8.0, 123456789, literal=bool(os.environ.get("POLEE_PREP_LITERAL")))
colptr, rowval, nzval = synth.to_csc(smp)
eff = smp["effective_lengths"]
ctx = P.Context(0)
for rep in range(int(os.environ.get("POLEE_PREP_REPS", "3"))):
    t = [time.time()]
    parents, js = (P.hclust(m, n, colptr, rowval, device=True, ctx=ctx) if os.environ.get("POLEE_PREP_TREE") == "cluster_device"
                   else P.hclust(m, n, colptr, rowval, parallel=True)); t.append(time.time())
    s = P.RNASeqSample(m, n, colptr, rowval, nzval, eff, ctx=ctx); t.append(time.time())
    tr = P.PolyaTreeTransform(parents, js, ctx=ctx); t.append(time.time())
    fit = P.LikelihoodApproximationFit(s, tr, num_steps=500, num_mc_samples=6, seed=rep); t.append(time.time())
    fit.run(500); fit.sync(); t.append(time.time())
    mu = fit.params(); del fit, tr, s; t.append(time.time())
    d = np.diff(t)
    print("tree %.3f  sample(layout+upload) %.3f  ptt %.3f  vi_create %.3f  fit %.3f  params+free %.3f  | total %.3f s"
          % (tuple(d) + (d.sum(),)), flush=True)
