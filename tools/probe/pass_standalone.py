"""The sparse pass ALONE (no VI loop around it): C2 sample of the gene-patterns generator, polee_loglik_eval 40 times.  Run under
rocprofv3 --kernel-trace --stats from the root of the tree whose library is to be measured (imports polee_amd and tools from the
current directory): the trace's durations of loglik_stream_kernel are the measurement.   usage: pass_standalone.py [literal]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import polee_amd as P
from tools import synth
literal = len(sys.argv) > 1 and sys.argv[1] == "literal"
n, m = 200000, 30000000
kw = dict(literal=True) if literal else {}
smp = synth.make_sample(n, m, 8.0, 123456789, **kw)
ctx = P.Context(0)
s = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
x = np.random.default_rng(3).dirichlet(np.ones(n), size=6).astype(np.float32)
for _ in range(40):
    s.log_likelihood(x, gradonly=True)
print("done", s.info["num_tiles"])
