"""Host layout builder against the device builder (stage mask = argv[1], default 4), case by case: which arrays differ.
argv[2] = "c2" adds the C2-size inputs with timings."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import polee_amd as P
from polee_amd import _lib as L
from tools import synth
from tools.probe import layout_hash as H


def build(ctx, smp, ks, mask):
    colptr, rowval, nzval = synth.to_csc(smp)
    m, n = int(smp["m"]), int(smp["n"])
    h = C.c_void_p()
    ksp = L.ptr(ks, L.i64p) if ks is not None else None
    t0 = time.time()
    if mask < 0:
        L.check(L.lib().polee_debug_psell_build(C.c_int64(m), C.c_int64(n), colptr.ctypes.data_as(C.c_void_p), colptr.dtype.itemsize,
                                                L.ptr(rowval, L.u32p), L.ptr(nzval, L.f32p), ksp, C.byref(h)))
    else:
        L.check(L.lib().polee_debug_psell_build_device(ctx._h, C.c_int64(m), C.c_int64(n), colptr.ctypes.data_as(C.c_void_p),
                                                       colptr.dtype.itemsize, L.ptr(rowval, L.u32p), L.ptr(nzval, L.f32p), ksp,
                                                       C.c_int(mask), C.byref(h)), ctx._h)
    dt = time.time() - t0
    v = L.PsellView()
    L.check(L.lib().polee_debug_psell_view(h, C.byref(v)))
    out = H.view_hashes(v)
    L.lib().polee_debug_psell_free(h)
    return out, dt


def main():
    mask = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ctx = P.Context()
    cases = list(H.cases())
    if len(sys.argv) > 2 and sys.argv[2] == "c2":
        cases = [("c2 patterns", synth.make_sample(200000, 20000000, 8.0, 123456789), None),
                 ("c2 literal", synth.make_sample(200000, 20000000, 8.0, 123456789, literal=True), None)]
    bad = 0
    for name, smp, ks in cases:
        a, ta = build(ctx, smp, ks, -1)
        b, tb = build(ctx, smp, ks, mask)
        diff = [k for k in a if a[k] != b[k] and not (k == "single_logsum" and abs(a[k] - b[k]) <= 1e-12 * abs(a[k]))]
        print("%-12s host %.2f s  device-mix %.2f s  %s" % (name, ta, tb, "IDENTICAL" if not diff else "DIFFERENT: " + ", ".join(diff)), flush=True)
        for k in diff:
            if not isinstance(a[k], str):
                print("    ", k, a[k], b[k])
        bad += bool(diff)
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
