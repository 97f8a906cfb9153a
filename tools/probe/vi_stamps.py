"""Phase stamps of the VI loop's tree kernels at C2 (diagnostic build: tools/probe/vi_stamps.sh builds libpolee_hip_vistamps.so
with -DPOLEE_VI_STAMPS and runs this through POLEE_HIP_LIB).  Thread 0 of every workgroup reads the 100 MHz clock at the
kernel's phase boundaries; printed: when the workgroups start and end relative to the first one, and the median time between
stamps."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import polee_amd as P
from polee_amd import _lib as L
from tools import synth
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789, literal=True)
parents, js = synth.make_tree(smp["gene"], 123456789)
ctx = P.Context(0)
sample = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
tree = P.PolyaTreeTransform(parents, js, ctx=ctx)
fit = P.LikelihoodApproximationFit(sample, tree, num_steps=1000, num_mc_samples=6, seed=1)
fit.run(50)
ctx.synchronize()
buf = np.zeros((4, 2048, 8), np.uint64)
f = L.lib().polee_debug_vi_stamps
f.argtypes = [ctypes.c_void_p]
f.restype = ctypes.c_int
assert f(buf.ctypes.data) == 0
names = {0: ("fwd", ["entry", "loads arrived", "open prefix", "block scan", "leaves+stores", "part sums"]),
         1: ("bwd", ["entry", "leaves loaded", "wave scan", "P written", "barrier", "H stored"]),
         2: ("update", ["entry", "inputs loaded", "gradient", "adam", "", "", "sampled+stored"])}
for kern, (name, labels) in names.items():
    s = buf[kern].astype(np.int64)
    used = s[:, 0] > 0
    s = s[used]
    if not len(s):
        continue
    t0 = s[:, 0].min()
    last = max(i for i in range(8) if (s[:, i] > 0).any())
    print("%s: %d workgroups; starts spread over %.2f us; first start -> last end %.2f us" % (
        name, len(s), (s[:, 0].max() - t0) / 100.0, (s[:, last].max() - t0) / 100.0))
    prev = 0
    for i in range(1, last + 1):
        if not (s[:, i] > 0).any():
            continue
        d = (s[:, i] - s[:, prev]) / 100.0
        print("   %-16s median %.2f us  (p10 %.2f, p90 %.2f)" % (labels[i] if i < len(labels) else i, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
        prev = i
    print("   workgroup lifetime median %.2f us" % np.median((s[:, last] - s[:, 0]) / 100.0))
