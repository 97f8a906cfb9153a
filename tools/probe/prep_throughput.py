"""End-to-end preparation of a cohort of C2-size samples on one GPU: tree construction (host) + device layout build
(host) + 500-iteration fit (device) per sample, `workers` samples in flight.
usage: python tools/probe/prep_throughput.py [jobs] [workers ...]
  POLEE_PREP_TREE=cluster_parallel   the rounds variant of the tree heuristic (polee_hclust_parallel); cluster_device: the same tree
                                     built on the GPU; cluster_auto: on the host when its CPUs are idle, else on the GPU
  POLEE_PREP_PROCESSES=1             worker processes (approximate_likelihood_cohort_processes) instead of threads
                                     (approximate_likelihood_cohort); POLEE_PREP_HOST_THREADS threads per process
The three distinct samples are generated once and kept as .npy files under /tmp; a worker maps the one it is given
(what reading its own likelihood-matrix file would be, without the HDF5 decode)."""
import functools
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np

n, m = 200000, 30000000
LITERAL = bool(os.environ.get("POLEE_PREP_LITERAL"))  # every fragment its own subset (the bench's headline input) instead of gene patterns
DIR = os.environ.get("POLEE_PREP_DIR", "/tmp/polee_prep_samples" + ("_literal" if LITERAL else ""))


def load_one(s):
    d = os.path.join(DIR, "s%d" % (s % 3))
    a = [np.load(os.path.join(d, k + ".npy"), mmap_mode="r") for k in ("colptr", "rowval", "nzval", "efflen")]
    return (m, n, a[0], a[1], a[2], np.asarray(a[3]))


def main():
    import polee_amd as P
    from tools import synth
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    workers_list = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 12]
    t0 = time.time()
    for s in range(3):
        d = os.path.join(DIR, "s%d" % s)
        if os.path.exists(os.path.join(d, "efflen.npy")):
            continue
        os.makedirs(d, exist_ok=True)
        smp = synth.make_sample(n, m, 8.0, 123456789 + 7919 * s, literal=LITERAL)
        colptr, rowval, nzval = synth.to_csc(smp)
        for k, v in (("colptr", colptr), ("rowval", rowval), ("nzval", nzval), ("efflen", smp["effective_lengths"])):
            np.save(os.path.join(d, k + ".npy"), v)
        del smp, colptr, rowval, nzval
    print("3 distinct samples ready in %.1f s" % (time.time() - t0), flush=True)
    approx = P.LogitSkewNormalPTTApprox(os.environ.get("POLEE_PREP_TREE", "cluster"))
    procs = bool(os.environ.get("POLEE_PREP_PROCESSES"))
    ht = int(os.environ["POLEE_PREP_HOST_THREADS"]) if os.environ.get("POLEE_PREP_HOST_THREADS") else None
    print("tree method:", approx.treemethod, "| workers are", ("processes, %s host threads each" % (ht or "usable CPUs / processes")) if procs else "threads",
          flush=True)
    loaders = [functools.partial(load_one, i) for i in range(jobs)]
    for w in workers_list:
        t0 = time.time()
        if procs:
            out = P.approximate_likelihood_cohort_processes(approx, loaders, processes=w, host_threads=ht, num_steps=500)
        else:
            out = P.approximate_likelihood_cohort(approx, loaders, workers=w, num_steps=500)
        dt = time.time() - t0
        ok = all(np.isfinite(o["mu"]).all() for o in out)
        print("workers %2d: %d samples in %.1f s = %.2f samples/s (%.2f s per sample), finite %s"
              % (w, jobs, dt, jobs / dt, dt / jobs, ok), flush=True)


if __name__ == "__main__":
    main()
