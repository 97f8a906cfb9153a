"""Where a wave of the uniform-stream part of loglik_fused_kernel spends its cycles (diagnostic build:
`make -C polee_amd/csrc EXTRA=-DPOLEE_STAMPS`, then run this on the GPU box; rebuild without the flag afterwards)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from polee_amd import _lib as L
from tools import synth
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789)
parents, js = synth.make_tree(smp["gene"], 123456789)
ctx = P.Context(0)
sample = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx,
                        xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
tree = P.PolyaTreeTransform(parents, js, ctx=ctx)
fit = P.LikelihoodApproximationFit(sample, tree, num_steps=40, num_mc_samples=6, seed=1)
fit.run(5); fit.sync()
out = (C.c_ulonglong * 16)()
f = L.lib().polee_debug_read_stamps
f(out)  # reset
fit.run(20); fit.sync()
f(out)
v = np.array(list(out), np.float64)
names = ["tile setup (dict, x window, barrier)", "slice bookkeeping", "waiting for the DMA", "run change: flush + column lookup",
         "phase 1 (LDS reads, MFMA, weights)", "phase 2 (MFMA)", "refill (DMA issue)", "final flush of the run",
         "waiting for the other waves", "global flush (atomics)"]
tot = v[:10].sum()
for nm, x in zip(names, v[:10]):
    print("%-42s %5.1f %%" % (nm, 100 * x / tot))
print("waves measured: %d, mean cycles per wave (memtime units): %.0f" % (v[11], v[10] / max(v[11], 1)))
