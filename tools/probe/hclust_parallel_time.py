"""polee_hclust_parallel on the C2 synthetic sample: wall time per call, the phase times (POLEE_BUILD_TIMING=1) and a hash
of the tree (the same for any POLEE_HOST_THREADS)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789)
colptr, rowval, nzval = synth.to_csc(smp)
for i in range(3):
    t0 = time.time()
    parents, js = P.hclust(m, n, colptr, rowval, parallel=True)
    print("parallel hclust %.3f s  tree %s" % (time.time() - t0, hashlib.sha1(parents.tobytes() + js.tobytes()).hexdigest()[:12]), flush=True)
if "--exact" in sys.argv:
    t0 = time.time()
    P.hclust(m, n, colptr, rowval)
    print("exact hclust %.3f s" % (time.time() - t0), flush=True)
