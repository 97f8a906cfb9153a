"""polee_loglik_create at C2 (device builder), patterns and literal: wall time of the call, best and median of POLEE_PREP_REPS, with the
builders' own phase times (POLEE_BUILD_TIMING=1) of the last repetition on stderr."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n, m = 200000, int(os.environ.get("POLEE_PREP_M", "30000000"))
reps = int(os.environ.get("POLEE_PREP_REPS", "5"))
ctx = P.Context(0)
for literal in (False, True):
    smp = synth.make_sample(n, m, 8.0, 123456789, literal=literal)
    colptr, rowval, nzval = synth.to_csc(smp)
    ts = []
    for rep in range(reps + 1):
        t0 = time.time()
        s = P.RNASeqSample(m, n, colptr, rowval, nzval, ctx=ctx)
        dt = time.time() - t0
        assert s.built_on_device
        del s
        if rep:
            ts.append(dt)
    a = np.array(ts)
    print("%-9s polee_loglik_create: best %.4f s  median %.4f s  (%d reps)" % ("literal" if literal else "patterns", a.min(), np.median(a), reps), flush=True)
