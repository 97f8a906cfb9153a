"""Where a wave of loglik_stream_kernel spends its cycles (diagnostic build:
`make -C polee_amd/csrc EXTRA=-DPOLEE_STAMPS`, then run this on the GPU box; rebuild without the flag afterwards)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from polee_amd import _lib as L
from tools import synth
# usage: stamps.py [mean nnz per fragment] [literal|patterns] [m]   (mean nnz 20 + literal: almost every slice is a wide one)
mean_nnz = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
literal = len(sys.argv) > 2 and sys.argv[2] == "literal"
n, m = 200000, int(sys.argv[3]) if len(sys.argv) > 3 else 30000000
smp = synth.make_sample(n, m, mean_nnz, 123456789, literal=literal)
parents, js = synth.make_tree(smp["gene"], 123456789)
ctx = P.Context(0)
sample = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx,
                        xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
tree = P.PolyaTreeTransform(parents, js, ctx=ctx)
fit = P.LikelihoodApproximationFit(sample, tree, num_steps=40, num_mc_samples=6, seed=1, profile=True)
fit.run(5); fit.sync()
out = (C.c_ulonglong * 24)()
f = L.lib().polee_debug_read_stamps
f(out)  # reset
fit.run(20); fit.sync()
f(out)
v = np.array(list(out), np.float64)
names = {0: "prefetch issue for the next tile", 1: "slice bookkeeping", 2: "waiting for the DMA",
         3: "run change: flush + column lookup", 11: "operand reads issued and landed (lgkmcnt 0)", 6: "ring refill (DMA issue)",
         14: "phase 1 MFMAs", 4: "weights", 5: "phase 2 (MFMA)",
         12: "mixed tile: the two sweeps", 7: "end of the wave's slices: run flush + queue drain",
         13: "next ring started (before / after the barrier)", 8: "waiting for the other waves (barrier A)",
         15: "tile flush issue (LDS -> global atomics)", 9: "barrier B", 10: "kernel tail (lp)"}
tot = v[:16].sum()
for i, nm in names.items():
    print("%-52s %5.1f %%" % (nm, 100 * v[i] / tot))
info = sample.info
print("input: mean nnz %.1f, %s; shares of nnz %s; kernel %.4f ms" % (mean_nnz, "literal" if literal else "patterns",
      [round(x / info["nnz"], 3) for x in info["stream_nnz"]], fit.stats()["loglik_kernel_ms_avg"]))
print("waves: %d, tiles per wave %.1f, slices per wave %.1f, mean cycles per wave (memtime units): %.0f"
      % (v[18], v[17] / max(v[18], 1), v[16] / max(v[18], 1), tot / max(v[18], 1)))
