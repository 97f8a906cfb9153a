"""sample_and_tree at C2 (tree on the device beside the layout build), one device copy of X (the default) against one upload per
builder (POLEE_SHARED_X=0), alternating; wall time per prepared sample, best and median of POLEE_PREP_REPS.  VERDICT r4 item 8."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n, m = 200000, int(os.environ.get("POLEE_PREP_M", "30000000"))
reps = int(os.environ.get("POLEE_PREP_REPS", "5"))
ctx = P.Context(0)
dev = P.LogitSkewNormalPTTApprox("cluster_device")
for literal in (False, True):
    smp = synth.make_sample(n, m, 8.0, 123456789, literal=literal)
    colptr, rowval, nzval = synth.to_csc(smp)
    eff = smp["effective_lengths"]
    times = {"1": [], "0": []}
    for rep in range(reps + 1):
        for mode in ("1", "0"):
            os.environ["POLEE_SHARED_X"] = mode
            t0 = time.time()
            s, t = P.sample_and_tree(dev, m, n, colptr, rowval, nzval, eff, ctx=ctx)
            dt = time.time() - t0
            del s, t
            if rep:  # (the first round warms the device cache)
                times[mode].append(dt)
    for mode, label in (("1", "one shared copy of X"), ("0", "one upload per builder")):
        a = np.array(times[mode])
        print("%-9s %-24s sample_and_tree: best %.3f s  median %.3f s  (%d reps)" % ("literal" if literal else "patterns", label, a.min(), np.median(a), reps), flush=True)
