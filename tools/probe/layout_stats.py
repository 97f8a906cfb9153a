"""Host-only statistics of the PSELL device layout of a synthetic sample (no GPU): tiles per stream, slices and
dictionary entries per tile, run lengths -- the quantities that size the fused kernel's prologue / flush amortisation.
usage: python tools/probe/layout_stats.py [c2|small|c1]"""
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
n, m, mean = W[sys.argv[1] if len(sys.argv) > 1 else "small"]
smp = synth.make_sample(n, m, mean, seed=123456789)
colptr, rowval, nzval = synth.to_csc(smp)
t0 = time.time()
h = C.c_void_p()
L.check(L.lib().polee_debug_psell_build(C.c_int64(m), C.c_int64(n), colptr.ctypes.data_as(C.c_void_p), 8,
                                        L.ptr(rowval, L.u32p), L.ptr(nzval, L.f32p), None, C.byref(h)))
print("build %.1f s" % (time.time() - t0))
v = L.PsellView()
L.check(L.lib().polee_debug_psell_view(h, C.byref(v)))
raw = np.ctypeslib.as_array(v.slice_off, shape=(v.num_slices + 1,))
off = (raw & np.uint32(0x1FFFFFFF)).astype(np.int64)
flags = (raw >> np.uint32(30))[:-1]
ts = np.ctypeslib.as_array(v.tile_slice, shape=(v.num_tiles + 1,)).astype(np.int64)
td = np.ctypeslib.as_array(v.tile_dict, shape=(v.num_tiles + 1,)).astype(np.int64)
ta1, ta = int(v.num_tiles_a1), int(v.num_tiles_a)
print("tiles", v.num_tiles, "A1", ta1, "A2", ta - ta1, "B", v.num_tiles - ta, "slices", v.num_slices, "dict", v.dict_len,
      "data MB", v.data_bytes / 1e6)
for name, a, b in (("A1", 0, ta1), ("A2", ta1, ta), ("B", ta, int(v.num_tiles))):
    if b <= a:
        continue
    ns = ts[a + 1:b + 1] - ts[a:b]
    nd = td[a + 1:b + 1] - td[a:b]
    by = (off[ts[a + 1:b + 1]] - off[ts[a:b]]) * 128
    q = lambda x: np.percentile(x, [5, 25, 50, 75, 95]).round(1).tolist()
    print(name, "slices/tile mean %.1f pct %s | dict/tile mean %.1f pct %s | KB/tile mean %.1f pct %s | total MB %.1f"
          % (ns.mean(), q(ns), nd.mean(), q(nd), by.mean() / 1e3, q(by / 1e3), by.sum() / 1e6))
    if name != "B":
        f = flags[ts[a]:ts[b]]
        cont = (f & 2) != 0
        nruns = int((~cont).sum())
        print("   runs %d, slices/run mean %.2f; runs/tile %.1f" % (nruns, len(f) / max(nruns, 1), nruns / (b - a)))
        # how many slices in tiles cut short by the dictionary cap (dict >= 240)?
        cut = nd >= 240
        print("   tiles cut by the dictionary cap: %.1f %% holding %.1f %% of the bytes; tiles with 64 slices %.1f %%"
              % (100 * cut.mean(), 100 * by[cut].sum() / by.sum(), 100 * (ns == 64).mean()))
        ww = (off[ts[a] + 1:ts[b] + 1] - off[ts[a]:ts[b]]) // 2 - 1
        print("   w: mean %.2f pct %s" % (ww.mean(), q(ww)))
L.lib().polee_debug_psell_free(h)
