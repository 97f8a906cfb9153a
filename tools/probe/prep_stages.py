"""Where a worker of the cohort pipeline spends its time at different degrees of concurrency (C2-size samples)."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
from concurrent.futures import ThreadPoolExecutor
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789)
colptr, rowval, nzval = synth.to_csc(smp)
eff = smp["effective_lengths"]
def job(i):
    t = [time.time()]
    parents, js = P.hclust(m, n, colptr, rowval); t.append(time.time())
    ctx = P.Context(0)
    s = P.RNASeqSample(m, n, colptr, rowval, nzval, eff, ctx=ctx); t.append(time.time())
    tr = P.PolyaTreeTransform(parents, js, ctx=ctx); t.append(time.time())
    fit = P.LikelihoodApproximationFit(s, tr, num_steps=500, num_mc_samples=6, seed=i); t.append(time.time())
    fit.run(500); fit.sync(); t.append(time.time())
    mu = fit.params(); del fit, tr, s; t.append(time.time())
    return np.diff(t)
for w in [int(a) for a in sys.argv[1:]] or [1, 12]:
    t0 = time.time()
    with ThreadPoolExecutor(w) as ex:
        d = np.array(list(ex.map(job, range(max(w, 2)))))
    print("workers %2d: wall %.1f s for %d samples; mean stage seconds: hclust %.2f, layout+upload %.2f, tree %.2f, vi_create %.2f, fit %.2f, params+free %.2f"
          % ((w, time.time() - t0, len(d)) + tuple(d.mean(axis=0))), flush=True)
