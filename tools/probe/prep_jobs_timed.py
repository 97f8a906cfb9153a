"""The cohort's job, one sample after the other, with the time of each step (needs the samples of prep_throughput.py under /tmp)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from polee_amd import core
from tools.probe import prep_throughput as PT

approx = P.LogitSkewNormalPTTApprox(os.environ.get("POLEE_PREP_TREE", "cluster_device"))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    t0 = time.time()
    m, n, colptr, rowval, nzval, eff = PT.load_one(i)
    t1 = time.time()
    ctx = core.Context(0)
    sample, tree = core.sample_and_tree(approx, m, n, colptr, rowval, nzval, eff, ctx=ctx)
    t2 = time.time()
    params = core.approximate_likelihood(approx, sample, tree, num_steps=500)
    t3 = time.time()
    del sample, tree, ctx
    t4 = time.time()
    print("job %2d: load %.3f  sample+tree %.3f  fit %.3f  free %.3f | %.3f s" % (i, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0), flush=True)
