"""Random small matrices of many shapes through both device builders and their host checkers: layouts byte for byte
(polee_debug_psell_build_device, all stages on the device), trees node for node (polee_hclust_parallel_device).
usage: fuzz_device_builders.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import polee_amd as P
from tools.probe import device_build_check as D


def sample_from_rows(rows, n, rng):
    lens = np.array([len(r) for r in rows], np.int64)
    tcolptr = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.uint64)
    trowval = (np.concatenate(rows) + 1).astype(np.uint32) if len(rows) and lens.sum() else np.zeros(0, np.uint32)
    tnzval = rng.uniform(1e-7, 1e-2, size=trowval.size).astype(np.float32)
    return dict(m=len(rows), n=n, nnz=int(trowval.size), tcolptr=tcolptr, trowval=trowval, tnzval=tnzval)


def random_case(rng):
    kind = rng.integers(0, 6)
    n = int(rng.integers(1, 400)) if rng.random() < 0.3 else int(rng.integers(400, 6000))
    m = int(rng.integers(0, 300)) if rng.random() < 0.15 else int(rng.integers(300, 60000))
    rows = []
    if kind == 0:  # unstructured
        hi = int(min(n, rng.integers(1, 40)))
        for _ in range(m):
            rows.append(np.sort(rng.choice(n, int(rng.integers(0, hi + 1)), replace=False)))
    else:  # genes of g isoforms; rows = a gene's pattern, a random subset of its isoforms, or a subset with strays from a neighbour
        gmax = [4, 12, 20, 40, 70][kind - 1]
        starts, p = [], 0
        while p < n:
            g = int(rng.integers(1, gmax + 1))
            starts.append((p, min(n, p + g)))
            p += g
        npat = int(rng.integers(1, 6))
        pats = {}
        weights = rng.gamma(0.5, size=len(starts)) + 1e-3
        genes = rng.choice(len(starts), size=m, p=weights / weights.sum())
        mode = rng.integers(0, 3)
        for gi in genes:
            a, b = starts[gi]
            if mode == 0 or (mode == 2 and rng.random() < 0.5):
                key = (gi, int(rng.integers(0, npat)))
                if key not in pats:
                    k = int(rng.integers(1, b - a + 1))
                    pats[key] = np.sort(rng.choice(np.arange(a, b), k, replace=False))
                r = pats[key]
            else:
                k = int(rng.integers(1, b - a + 1))
                r = np.sort(rng.choice(np.arange(a, b), k, replace=False))
            if rng.random() < 0.03 and gi + 1 < len(starts):
                c, d = starts[gi + 1]
                r = np.unique(np.concatenate([r, rng.choice(np.arange(c, d), 1)]))
            if rng.random() < 0.01:
                r = r[:0]
            rows.append(r)
        if rng.random() < 0.3:  # fragment order: sorted by gene or shuffled
            rng.shuffle(rows)
    smp = sample_from_rows(rows, n, rng)
    ks = rng.integers(1, 9, size=smp["m"]).astype(np.int64) if rng.random() < 0.3 else None
    return kind, smp, ks


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = P.Context()
    from tools import synth
    bad = 0
    for i in range(cases):
        kind, smp, ks = random_case(rng)
        try:
            a, _ = D.build(ctx, smp, ks, -1)
            b, _ = D.build(ctx, smp, ks, 7)
            diff = [k for k in a if a[k] != b[k] and not (k == "single_logsum" and abs(a[k] - b[k]) <= 1e-12 * max(1.0, abs(a[k])))]
        except Exception as e:  # both builders must refuse the same inputs; none of these is malformed
            diff = ["exception: %s" % e]
        tdiff = ""
        if smp["n"] >= 1 and smp["nnz"] > 0:
            colptr, rowval, _ = synth.to_csc(smp)
            ph, jh = P.hclust(smp["m"], smp["n"], colptr, rowval, parallel=True)
            pd, jd = P.hclust(smp["m"], smp["n"], colptr, rowval, device=True, ctx=ctx)
            if not (np.array_equal(ph, pd) and np.array_equal(jh, jd)):
                tdiff = "TREE DIFFERS"
        if diff or tdiff:
            bad += 1
            print("case %d kind %d n %d m %d nnz %d ks %s: %s %s" % (i, kind, smp["n"], smp["m"], smp["nnz"], ks is not None, diff, tdiff), flush=True)
    print("cases %d, mismatching %d" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
