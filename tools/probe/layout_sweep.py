"""Host-only sensitivity of the PSELL device layout to SET DIVERSITY (no GPU): stored bytes per non-zero against CSR's
for the synthetic generator as built, with per-entry dropout p, with every fragment drawing its own subset ("literal",
SURVEY 8(d)'s wording), and for the reference fixture tiled block-diagonally.
usage: python tools/probe/layout_sweep.py [small|c2] [case ...]      cases: p0 p0.1 p0.3 literal fixture"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polee_amd import _lib as L  # noqa: E402
from tools import synth  # noqa: E402

W = {"c1": (1000, 100000, 2.2), "small": (20000, 3000000, 8.0), "c2": (200000, 30000000, 8.0)}


def build_stats(smp):
    m, n = smp["m"], smp["n"]
    colptr, rowval, nzval = synth.to_csc(smp)
    t0 = time.time()
    h = C.c_void_p()
    L.check(L.lib().polee_debug_psell_build(C.c_int64(m), C.c_int64(n), colptr.ctypes.data_as(C.c_void_p), 8,
                                            L.ptr(rowval, L.u32p), L.ptr(nzval, L.f32p), None, C.byref(h)))
    dt = time.time() - t0
    v = L.PsellView()
    L.check(L.lib().polee_debug_psell_view(h, C.byref(v)))
    nnz = int(v.nnz)
    out = dict(nnz_per_row=nnz / m, stored_entries_per_nnz=v.padded_nnz / nnz,
               stored_bytes_per_nnz=(v.data_bytes + 4 * (v.num_slices + 1) + 4 * v.dict_len + v.stream_bytes[6]) / nnz,
               csr_bytes_per_nnz=(8 * nnz + 4 * (m + 1)) / nnz, build_s=dt, tiles=int(v.num_tiles),
               tiles_a=int(v.num_tiles_a), tiles_a1=int(v.num_tiles_a1),
               share=[v.stream_nnz[i] / nnz for i in range(6)],
               bpn=[v.stream_bytes[i] / max(v.stream_nnz[i], 1) for i in range(6)])
    L.lib().polee_debug_psell_free(h)
    return out


def cases(wl, names):
    n, m, mean = W[wl]
    for c in names:
        if c == "fixture":
            reps = max(1, round(n / 313))
            yield "tiled fixture x%d" % reps, synth.tile_fixture(reps)
        elif c == "literal":
            yield "literal subsets", synth.make_sample(n, m, mean, seed=123456789, literal=True)
        else:
            p = float(c[1:])
            yield "dropout p=%g" % p, synth.make_sample(n, m, mean, seed=123456789, dropout=p)


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "small"
    names = sys.argv[2:] or ["p0", "p0.1", "p0.3", "literal", "fixture"]
    print("| input | nnz/row | stored entries / nnz | stored bytes / nnz | CSR bytes / nnz | build s | share of nnz: dense<=16 / masked<=16 / dense 17..32 / masked 17..32 / mixed<=15 / mixed (2nd launch) | slice bytes / nnz per stream |")
    print("|---|---|---|---|---|---|---|---|")
    for name, smp in cases(wl, names):
        s = build_stats(smp)
        print("| %s | %.2f | %.3f | **%.2f** | %.2f | %.2f | %s | %s |" % (
            name, s["nnz_per_row"], s["stored_entries_per_nnz"], s["stored_bytes_per_nnz"], s["csr_bytes_per_nnz"],
            s["build_s"], " / ".join("%.3f" % x for x in s["share"]), " / ".join("%.2f" % x for x in s["bpn"])), flush=True)
