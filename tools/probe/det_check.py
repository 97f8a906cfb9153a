"""Deterministic mode: how often do repeated passes over the same inputs differ, and where?  (fixture-sized and tiled inputs)
usage: det_check.py [reps of the fixture, default 1] [evaluations, default 30]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 30
smp = synth.tile_fixture(reps)
n, m = smp["n"], smp["m"]
ctx = P.Context(0)
s = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
info = s.info
print("tiles per stream", info["stream_tiles"], "nnz shares", [round(v / info["nnz"], 4) for v in info["stream_nnz"]])
x = np.random.default_rng(3).dirichlet(np.ones(n), size=6).astype(np.float32)
if os.environ.get("DET_DIRTY"):  # fill the block cache with NaN-patterned blocks first (what earlier work in a process leaves behind)
    import ctypes as C
    from polee_amd import core
    junk = np.full(1 << 22, np.nan, np.float32)
    for k in range(6):
        c2 = P.Comm(ctx, 1, 0)
        c2.allreduce_sum(junk[: (1 << 22) >> k])  # (uploads into a DevBuf of that size class and releases it)
    lp0, g0 = s.log_likelihood(x)
s.set_deterministic(True)
lp1, g1 = s.log_likelihood(x)
print("NaNs in the deterministic result: lp %d, gradient %d" % (int(np.isnan(lp1).sum()), int(np.isnan(g1).sum())))
bad_lp = bad_g = 0
cols = set()
for _ in range(evals):
    lp2, g2 = s.log_likelihood(x)
    bad_lp += int(not np.array_equal(lp1, lp2))
    d = np.argwhere(g1 != g2)
    bad_g += int(len(d) > 0)
    cols |= set(d[:, 1].tolist())
print("evaluations %d: lp differs in %d, gradient differs in %d; transcripts involved: %d %s" % (evals, bad_lp, bad_g, len(cols), sorted(cols)[:12]))
if cols:
    # which sets do those transcripts sit in?
    ptr, col = smp["tcolptr"].astype(np.int64) - 1, smp["trowval"].astype(np.int64) - 1
    lens = np.diff(ptr)
    for c in sorted(cols)[:4]:
        rows = np.unique(np.searchsorted(ptr, np.flatnonzero(col == c), side="right") - 1)
        print("  transcript %d: in %d fragments, row lengths %s" % (c, len(rows), np.bincount(lens[rows])[:40].tolist()))
