"""Per-tile and per-workgroup cycle counts of loglik_stream_kernel (diagnostic build, see run_stamps.sh): how well does
the static schedule balance the workgroups, and what does a tile of each stream cost per byte?
usage (GPU box): tools/probe/run_stamps.sh is for stamps.py; this one: make EXTRA=-DPOLEE_STAMPS, then
python3 tools/probe/tile_cycles.py [dropout]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import polee_amd as P
from polee_amd import _lib as L
from tools import synth
n, m = 200000, 30000000
# argument: a per-entry dropout probability, "literal" (every fragment its own subset) or "fixture" (the real fixture tiled x639)
mode = sys.argv[1] if len(sys.argv) > 1 else "0"
if mode == "fixture":
    smp = synth.tile_fixture(639)
    n, m = smp["n"], smp["m"]
elif mode == "literal":
    smp = synth.make_sample(n, m, 8.0, 123456789, literal=True)
else:
    smp = synth.make_sample(n, m, 8.0, 123456789, dropout=float(mode))
parents, js = synth.make_tree(smp["gene"], 123456789)
ctx = P.Context(0)
sample = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx,
                        xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
info = sample.info
tree = P.PolyaTreeTransform(parents, js, ctx=ctx)
fit = P.LikelihoodApproximationFit(sample, tree, num_steps=80, num_mc_samples=6, seed=1, profile=True)
fit.run(20); fit.sync()
NT = sum(info["stream_tiles"][:5])
tiles = (C.c_ulonglong * NT)(); wgs = (C.c_ulonglong * 1024)()
f = L.lib().polee_debug_read_tile_cycles
f(tiles, NT, wgs, 1024)  # reset
R = 40
fit.run(R); fit.sync()
f(tiles, NT, wgs, 1024)
t = np.array(list(tiles), np.float64) / R
w = np.array(list(wgs), np.float64) / R
w = w[w > 0]
print("workgroups %d: cycles mean %.0f  max %.0f  min %.0f  (max/mean %.3f)  p95 %.0f" % (len(w), w.mean(), w.max(), w.min(), w.max() / w.mean(), np.percentile(w, 95)))
ends = np.cumsum([0] + info["stream_tiles"][:5])
bytes_ = info["stream_bytes_hbm"]
for i, name in enumerate(["A1", "A1M", "A2", "A2M", "BN"]):
    a, b = ends[i], ends[i + 1]
    if b <= a:
        continue
    tt = t[a:b]
    print("%-4s tiles %6d  cycles/tile mean %8.0f (p5 %6.0f p95 %6.0f)  total %5.1f %% of tile time  cycles per KiB %.1f  share of nnz %.3f  cycles per nnz %.2f" % (
        name, b - a, tt.mean(), np.percentile(tt, 5), np.percentile(tt, 95), 100 * tt.sum() / t.sum(), tt.sum() / (bytes_[i] / 1024),
        info["stream_nnz"][i] / info["nnz"], tt.sum() / max(info["stream_nnz"][i], 1)))

st = fit.stats()
print("mode %s: kernel ms %.4f  pass ms %.4f  single-transcript rows %d (%.3f of nnz)" % (
    mode, st["loglik_kernel_ms_avg"], st["loglik_pass_ms_avg"], info["stream_rows"][7], info["stream_nnz"][7] / info["nnz"]))
if os.environ.get("TILE_FEATURES") != "1":
    sys.exit(0)
# per-tile features of the layout (host build of the same matrix), saved beside the cycles for an offline fit of the
# schedule's cost model
colptr, rowval, nzval = synth.to_csc(smp)
h = C.c_void_p()
L.check(L.lib().polee_debug_psell_build(C.c_int64(m), C.c_int64(n), colptr.ctypes.data_as(C.c_void_p), 8,
                                        L.ptr(rowval, L.u32p), L.ptr(nzval, L.f32p), None, C.byref(h)))
v = L.PsellView()
L.check(L.lib().polee_debug_psell_view(h, C.byref(v)))
ts = np.ctypeslib.as_array(v.tile_slice, shape=(v.num_tiles + 1,)).astype(np.int64)
td = np.ctypeslib.as_array(v.tile_dict, shape=(v.num_tiles + 1,)).astype(np.int64)
sw = np.ctypeslib.as_array(v.slice_w, shape=(v.num_slices,)).astype(np.int64)
raw = np.ctypeslib.as_array(v.slice_off, shape=(v.num_slices + 1,))
off = (raw & np.uint32(0x1FFFFFFF)).astype(np.int64)
cont = ((raw >> np.uint32(30)) & 2)[:-1] != 0
ro = np.ctypeslib.as_array(v.row_order, shape=(v.num_slices * 64,)).reshape(-1, 64)
rows = (ro != 0xFFFFFFFF).sum(1)
assert v.num_tiles_s == NT, (v.num_tiles_s, NT)
kind = np.zeros(NT, np.int64)
for i in range(5):
    kind[ends[i]:ends[i + 1]] = i
sl_bytes = (off[1:] - off[:-1]) * 128
grp = (sw + 3) // 4
cs = lambda a: np.concatenate([[0], np.cumsum(a)])
feat = {}
for name, a in (("nsl", np.ones_like(sw)), ("bytes", sl_bytes), ("groups", grp), ("starts", (~cont).astype(np.int64)),
                ("start_groups", np.where(cont, 0, grp)), ("w", sw), ("rows", rows)):
    c = cs(a)
    feat[name] = c[ts[1:NT + 1]] - c[ts[:NT]]
feat["dict"] = td[1:NT + 1] - td[:NT]
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "tile_cycles_%s.npz" % os.environ.get("TAG", "x")),
         cycles=t, kind=kind, wg=w, **feat)
