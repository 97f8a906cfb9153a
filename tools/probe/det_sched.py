"""Deterministic mode, gradient-only passes: are repeated launches bitwise equal (dynamic tile schedule; POLEE_DET_STATIC=1: static lists)?"""
import sys, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from tools import synth
literal = os.environ.get("DET_LITERAL", "1") == "1"
smp = synth.make_sample(20000, 3000000, 8.0, 77, literal=literal)
c, r, v = synth.to_csc(smp)
ctx = P.Context(0)
s = P.RNASeqSample(smp["m"], smp["n"], c, r, v, ctx=ctx)
i = s.info
print("tiles", i["num_tiles"], "stream rows", i["stream_rows"], "stream tiles", i["stream_tiles"])
s.set_deterministic(True)
x = np.random.default_rng(5).dirichlet(np.ones(smp["n"]), size=6).astype(np.float32)
gs = []
for _ in range(8):
    lp, g = s.log_likelihood(x, gradonly=True)
    gs.append(g.copy())
hs = [hashlib.sha256(np.ascontiguousarray(g).tobytes()).hexdigest()[:12] for g in gs]
print("hashes", hs)
for g in gs[1:]:
    d = np.argwhere(g != gs[0])
    if len(d):
        print("differs at", len(d), "entries; first", d[:5].tolist(), "values", [(float(gs[0][tuple(q)]), float(g[tuple(q)])) for q in d[:3]])
        break
