"""Does tree construction scale over concurrent samples?  12 at once as threads of one process vs as forked processes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import multiprocessing as mp
from concurrent.futures import ThreadPoolExecutor
import polee_amd as P
from tools import synth
n, m = 200000, 30000000
smp = synth.make_sample(n, m, 8.0, 123456789)
colptr, rowval, nzval = synth.to_csc(smp)
def one(i):
    t0 = time.time(); P.hclust(m, n, colptr, rowval); return time.time() - t0
if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    print("solo: %.2f s" % one(0), flush=True)
    t0 = time.time()
    with ThreadPoolExecutor(W) as ex: d = list(ex.map(one, range(W)))
    print("%d threads : wall %.1f s, mean per call %.2f s" % (W, time.time() - t0, np.mean(d)), flush=True)
    t0 = time.time()
    with mp.get_context("fork").Pool(W) as pool: d = pool.map(one, range(W))
    print("%d processes: wall %.1f s, mean per call %.2f s" % (W, time.time() - t0, np.mean(d)), flush=True)
