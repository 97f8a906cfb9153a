"""The kept device buffers under their debug mode (POLEE_DEVICE_CACHE_POISON=1, csrc/common.hpp DevBlockCache):
every released block is filled with a pattern on its owner's stream and verified when it is handed out again or freed, so a kernel
that still writes to a block after its owner released it -- the suspected cause of the zeros seen under hipMallocAsync -- shows up.
Runs, in one process: the device builders' fuzz cases, a cohort of C2-size samples prepared by worker threads (device tree on
its own context beside the device layout build, then the fit), optionally one C5-size sample; prints the counters.
usage: POLEE_DEVICE_CACHE_POISON=1 python tools/probe/poison_run.py [fuzz cases] [cohort jobs] [workers] [c5: 0/1]"""
import functools
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np

assert os.environ.get("POLEE_DEVICE_CACHE_POISON") == "1", "set POLEE_DEVICE_CACHE_POISON=1"
import polee_amd as P
from polee_amd import core
from tools import synth
from tools.probe import fuzz_device_builders as F

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 850
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
with_c5 = len(sys.argv) > 4 and sys.argv[4] == "1"


def report(what):
    chk, bad, words = core.device_cache_poison_stats()
    print("%-40s blocks verified %8d  overwritten after release %d (%d words)  kept now %.2f GB" % (what, chk, bad, words, core.device_cache_bytes() / 1e9), flush=True)


t0 = time.time()
sys.argv = [sys.argv[0], str(cases), "1"]
rc = F.main()
report("fuzz cases (%d, rc %d, %.0f s)" % (cases, rc, time.time() - t0))


def make(seed, m, literal):
    smp = synth.make_sample(200000, m, 8.0, seed, literal=literal)
    colptr, rowval, nzval = synth.to_csc(smp)
    return (m, 200000, colptr, rowval, nzval, smp["effective_lengths"])


t0 = time.time()
S = [make(123456789, 30000000, False), make(123456789 + 7919, 30000000, True)]
approx = P.LogitSkewNormalPTTApprox("cluster_device")
out = P.approximate_likelihood_cohort(approx, [functools.partial(lambda i: S[i % 2], i) for i in range(jobs)], workers=workers, num_steps=200)
ok = all(np.isfinite(o["mu"]).all() for o in out)
report("C2 cohort (%d samples, %d workers, finite %s, %.0f s)" % (jobs, workers, ok, time.time() - t0))
del S, out
if with_c5:
    t0 = time.time()
    s5 = make(5, 150000000, True)
    out = P.approximate_likelihood_cohort(approx, [lambda: s5], workers=1, num_steps=50)
    report("C5 sample (finite %s, %.0f s)" % (bool(np.isfinite(out[0]["mu"]).all()), time.time() - t0))
    del s5, out
core.host_cache_trim()
report("after trim")
chk, bad, words = core.device_cache_poison_stats()
sys.exit(1 if bad or rc or chk == 0 else 0)
