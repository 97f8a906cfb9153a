"""The device tree builder against the host's rounds variant: the serialised trees must be the same arrays.
argv: "c2" adds the C2-size sample with timings."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import polee_amd as P
from tools import synth
from tools.probe import layout_hash as H


def main():
    ctx = P.Context()
    cases = [(name, smp) for name, smp, _ in H.cases()]
    if len(sys.argv) > 1 and sys.argv[1] == "c2":
        cases = [("c2 patterns", synth.make_sample(200000, 30000000, 8.0, 123456789))]
    bad = 0
    for name, smp in cases:
        colptr, rowval, _ = synth.to_csc(smp)
        t0 = time.time()
        ph, jh = P.hclust(smp["m"], smp["n"], colptr, rowval, parallel=True)
        t1 = time.time()
        pd, jd = P.hclust(smp["m"], smp["n"], colptr, rowval, device=True, ctx=ctx)
        t2 = time.time()
        if len(sys.argv) > 1:
            pd, jd = P.hclust(smp["m"], smp["n"], colptr, rowval, device=True, ctx=ctx)
            t3 = time.time()
            t2 = t1 + (t3 - t2)
        same = np.array_equal(ph, pd) and np.array_equal(jh, jd)
        first = int(np.argmax((ph != pd) | (jh != jd))) if not same else -1
        print("%-12s n=%d host %.3f s  device %.3f s  %s" % (name, smp["n"], t1 - t0, t2 - t1, "IDENTICAL" if same else "DIFFERENT from position %d" % first), flush=True)
        bad += not same
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
