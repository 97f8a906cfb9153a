"""Phase times of polee_hclust on the C2 synthetic sample (POLEE_BUILD_TIMING=1 prints them)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789)
colptr, rowval, nzval = synth.to_csc(smp)
t0 = time.time()
parents, js = P.hclust(m, n, colptr, rowval)
print("hclust %.2f s" % (time.time() - t0), parents[:5], js[:5])
