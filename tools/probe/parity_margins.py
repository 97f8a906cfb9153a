"""the margin of the f32-accumulating sparse pass under the north star's 1e-4, on THIS build, in the default
(float-atomic) mode whose summation order changes from run to run.  For C2 literal, C2 patterns and C5 literal: K = 6 draws
evaluated five times each through the C ABI; against the oracle's f64 accumulation (oracle/polee_oracle.c log_likelihood,
sparse.jl:13-17,32-36): the worst weighted gradient error over all transcripts and repetitions
(|g - g_ref| / (|g_ref| + 1e-2 max|g_ref|), the measure of tests/test_gpu_configs.py), the relative error on the five longest
columns (the most f32 additions into one sum), the relative lp error, and the spread between repetitions.  Exit code 1 when any
figure is within 10x of 1e-4 (margin below 10) -- lp is held to the tests' 1e-6.

usage: parity_margins.py [c2_literal c2_patterns c5_literal]     (writes to stdout; tools/probe/parity_margins.sh tees it)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import polee_amd as P
from oracle import oracle as O
from tools import synth

CASES = {"c2_literal": (200_000, 30_000_000, dict(literal=True)), "c2_patterns": (200_000, 30_000_000, dict()),
         "c5_literal": (200_000, 150_000_000, dict(literal=True))}
REPS, K, TOL, WANT = 5, 6, 1e-4, 10.0
bad = False
print("build: %s" % P.version())
for name in (sys.argv[1:] or list(CASES)):
    n, m, kw = CASES[name]
    t0 = time.time()
    smp = synth.make_sample(n, m, 8.0, seed=123456789, **kw)
    ctx = P.Context(0)
    s = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    rng = np.random.default_rng(0)
    x = rng.gamma(0.3, size=(K, n)).astype(np.float32) + np.float32(1e-7)
    x /= x.sum(axis=1, keepdims=True)
    x = np.clip(x, np.float32(1e-10), 1)
    runs = [s.log_likelihood(x) for _ in range(REPS)]  # (lp [K], g [K][n]) per repetition
    del s
    # the oracle on blocks of rows (C5's CSC does not fit beside everything else in one piece)
    O.set_num_threads(O.physical_cores())
    lpo, go = np.zeros(K), np.zeros((K, n))
    nblk = 5 if m > 50_000_000 else 1
    collen = np.zeros(n, np.int64)
    for b in range(nblk):
        r0, r1 = (m * b) // nblk, (m * (b + 1)) // nblk
        from polee_amd.cohort import take_rows
        bp, br, bv = take_rows(smp["tcolptr"], smp["trowval"], smp["tnzval"], r0, r1)
        colptr, rowval, nzval = synth.to_csc(dict(m=r1 - r0, n=n, nnz=int(len(br)), tcolptr=np.ascontiguousarray(bp), trowval=np.ascontiguousarray(br), tnzval=np.ascontiguousarray(bv)))
        collen += np.diff(colptr.astype(np.int64))
        so = O.Sample(r1 - r0, n, colptr, rowval, nzval)
        for k in range(K):
            l, g = so.log_likelihood(x[k])
            lpo[k] += l
            go[k] += g
        del so, colptr, rowval, nzval
    longest = np.argsort(collen)[-5:]
    worst, worst_at, lp_err, long_err = 0.0, None, 0.0, 0.0
    for r, (lp, g) in enumerate(runs):
        for k in range(K):
            scale = np.abs(go[k]).max()
            e = np.abs(g[k] - go[k]) / (np.abs(go[k]) + 1e-2 * scale)
            j = int(np.argmax(e))
            if e[j] > worst:
                worst, worst_at = float(e[j]), (r, k, j, int(collen[j]))
            lp_err = max(lp_err, abs(lp[k] - lpo[k]) / abs(lpo[k]))
            long_err = max(long_err, float((np.abs(g[k][longest] - go[k][longest]) / np.abs(go[k][longest])).max()))
    g_all = np.stack([g for _, g in runs])
    spread = float((np.ptp(g_all, axis=0) / (np.abs(go) + 1e-2 * np.abs(go).max(axis=1, keepdims=True))).max())
    margin = TOL / max(worst, long_err, 1e-300)
    print("%s: n %d, m %d, nnz %d, %d repetitions x %d draws (%.0f s)" % (name, n, m, smp["nnz"], REPS, K, time.time() - t0))
    print("   worst weighted gradient error %.3g (repetition %d, draw %d, transcript %d: a column of %d fragments)" % ((worst,) + worst_at))
    print("   five longest columns (%s fragments): worst relative error %.3g" % (collen[longest].tolist(), long_err))
    print("   lp: worst relative error %.3g;   run-to-run spread of the gradient (same measure) %.3g" % (lp_err, spread))
    print("   margin under 1e-4: %.0fx %s" % (margin, "" if margin >= WANT and lp_err <= 1e-6 else "  <-- BELOW 10x"))
    bad |= margin < WANT or lp_err > 1e-6
    del runs, g_all, smp
sys.exit(1 if bad else 0)
