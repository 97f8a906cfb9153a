"""End-to-end preparation of a cohort of C2-size samples on one GPU: tree construction (host) + device layout build
(host) + 500-iteration fit (device) per sample, `workers` samples in flight (polee_amd.approximate_likelihood_cohort).
usage: python tools/probe/prep_throughput.py [jobs] [workers ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
n, m = 200000, 30000000
jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
workers_list = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 12]
t0 = time.time()
distinct = []
for s in range(3):
    smp = synth.make_sample(n, m, 8.0, 123456789 + 7919 * s)
    colptr, rowval, nzval = synth.to_csc(smp)
    distinct.append((m, n, colptr, rowval, nzval, smp["effective_lengths"]))
    del smp
print("3 distinct samples generated in %.1f s" % (time.time() - t0), flush=True)
# POLEE_PREP_TREE=cluster_parallel: the rounds variant of the tree heuristic (polee_hclust_parallel)
approx = P.LogitSkewNormalPTTApprox(os.environ.get("POLEE_PREP_TREE", "cluster"))
print("tree method:", approx.treemethod, flush=True)
for w in workers_list:
    t0 = time.time()
    out = P.approximate_likelihood_cohort(approx, [distinct[i % 3] for i in range(jobs)], workers=w, num_steps=500)
    dt = time.time() - t0
    ok = all(np.isfinite(o["mu"]).all() for o in out)
    print("workers %2d: %d samples in %.1f s = %.2f samples/s (%.2f s per sample), finite %s" % (w, jobs, dt, jobs / dt, dt / jobs, ok), flush=True)
