import ctypes, sys, os, json, subprocess
sys.path.insert(0, '.')
import numpy as np
import polee_amd as P
from polee_amd import _lib
from tools import synth
wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
n, m, nn = {'small': (20000, 3000000, 8.0), 'c2': (200000, 30000000, 8.0)}[wl]
smp = synth.make_sample(n, m, nn, 123456789)
par, js = synth.make_tree(smp['gene'], 1)
ctx = P.Context(0)
s = P.RNASeqSample(m, n, None, None, None, smp['effective_lengths'], ctx=ctx, xt=(smp['tcolptr'], smp['trowval'], smp['tnzval']))
t = P.PolyaTreeTransform(par, js, ctx=ctx)
fit = P.LikelihoodApproximationFit(s, t, num_steps=30, num_mc_samples=6)
fit.run(5); fit.sync()
out = (ctypes.c_ulonglong * 16)()
L = _lib.lib()
L.polee_debug_read_stamps(out)
fit.run(10); fit.sync()
L.polee_debug_read_stamps(out)
v = np.array(list(out), dtype=np.float64)
names = ['prologue','bookkeeping','dma_wait','run_change','phase1','phase2','refill','final_flush','wg_wait','global_flush']
waves, slices = v[11], v[10]
print(wl, 'waves', waves, 'slices', slices, 'slices/wave', slices/waves)
tot = v[:10].sum()
for i, nm in enumerate(names):
    print('%-14s %8.0f cyc/wave  %6.0f cyc/slice  %5.1f%%' % (nm, v[i]/waves, v[i]/slices, 100*v[i]/tot))
print('total cycles per wave', tot/waves)
