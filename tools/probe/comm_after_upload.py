"""Does RCCL initialise after a large sample has been uploaded in the same process? (order dependence seen in the GPU suite)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, m = (200000, 30000000) if which == "c2" else (200000, 150000000)
ctx = P.Context(0)
smp = synth.make_sample(n, m, 8.0, 123456789)
s = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
print("uploaded", s.info["nnz"], flush=True)
if len(sys.argv) > 2:
    import torch
    print("torch sees", torch.cuda.device_count(), flush=True)
try:
    c = P.Comm(ctx, 1, 0)
    print("comm ok", c.allreduce_sum(np.ones(4, np.float32)))
except Exception as e:
    print("comm FAILED:", e)
