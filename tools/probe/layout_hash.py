"""Hashes of every array of a PSELL layout through the debug view (host builder, or the device builder with
--device): the two builders -- and a refactored builder and its predecessor -- must agree byte for byte."""
import sys, os, hashlib, json, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from polee_amd import _lib as L
from tools import synth


def view_hashes(v):
    def h(ptr, count, dtype):
        if not ptr or count == 0:
            return "-"
        a = np.ctypeslib.as_array(ptr, shape=(int(count),))
        return hashlib.sha1(np.ascontiguousarray(a).view(np.uint8)).hexdigest()[:16]
    ns, nt = int(v.num_slices), int(v.num_tiles)
    out = dict(scalars=[int(x) for x in (v.m, v.n, v.nnz, ns, nt, v.padded_nnz, v.num_empty_rows, v.data_bytes, v.dict_len,
                                         v.max_row_nnz, v.max_tile_cols, v.num_tiles_a, v.num_tiles_a1, v.num_tiles_a1m,
                                         v.num_tiles_a2, v.num_tiles_s, v.csr_num_rows, v.single_num_rows)],
               stream_rows=list(v.stream_rows), stream_nnz=list(v.stream_nnz), stream_bytes=list(v.stream_bytes),
               data=h(v.data, v.data_bytes, np.uint8), slice_off=h(v.slice_off, ns + 1, np.uint32),
               tile_slice=h(v.tile_slice, nt + 1, np.uint32), tile_dict=h(v.tile_dict, nt + 1, np.uint32),
               dict=h(v.dict, v.dict_len, np.uint32), row_order=h(v.row_order, ns * 64, np.uint32),
               slice_ks=h(v.slice_ks, ns * 64, np.float32), slice_flags=h(v.slice_flags, ns, np.uint8),
               slice_w=h(v.slice_w, ns, np.uint8), csr_rows=h(v.csr_rows, v.csr_num_rows, np.uint32),
               csr_rowptr=h(v.csr_rowptr, v.csr_num_rows + 1 if v.csr_num_rows else 0, np.uint32),
               single_rows=h(v.single_rows, v.single_num_rows, np.uint32), single_cnt=h(v.single_cnt, v.n if v.single_num_rows else 0, np.float32),
               single_logsum=float(v.single_logsum))
    return out


def random_rows(rng, m, n, lo, hi, local=None):
    """rows of lo..hi random transcripts (within a window of `local` ids, or anywhere): no structure for the builder"""
    lens = rng.integers(lo, hi + 1, size=m)
    tcolptr = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.uint64)
    cols = []
    for i in range(m):
        if local:
            base = int(rng.integers(0, n - local))
            c = base + rng.choice(local, lens[i], replace=False)
        else:
            c = rng.choice(n, lens[i], replace=False)
        cols.append(np.sort(c))
    trowval = (np.concatenate(cols) + 1).astype(np.uint32)
    tnzval = rng.uniform(1e-6, 1e-2, size=trowval.size).astype(np.float32)
    return dict(m=m, n=n, nnz=int(trowval.size), tcolptr=tcolptr, trowval=trowval, tnzval=tnzval)


def cases(scale=1):
    rng = np.random.default_rng(5)
    yield "patterns", synth.make_sample(20000, 600000 * scale, 8.0, 11), None
    yield "literal", synth.make_sample(20000, 400000 * scale, 8.0, 12, literal=True), None
    yield "dropout0.3", synth.make_sample(20000, 400000 * scale, 8.0, 13, dropout=0.3), None
    s = synth.make_sample(5000, 200000 * scale, 12.0, 14, dropout=0.1)
    yield "ks", s, rng.integers(1, 6, size=200000 * scale).astype(np.int64)
    yield "wide", synth.make_sample(3000, 150000 * scale, 24.0, 15, dropout=0.2), None
    yield "random", random_rows(rng, 60000, 20000, 3, 10), None
    yield "long", random_rows(rng, 30000, 3000, 20, 150, local=400), None
    try:
        yield "fixture", synth.tile_fixture(3), None
    except Exception as e:  # (no golden directory)
        print("fixture skipped:", e, file=sys.stderr)


def build_host(smp, ks):
    colptr, rowval, nzval = synth.to_csc(smp)
    m, n = int(smp["m"]), int(smp["n"])
    h = C.c_void_p()
    L.check(L.lib().polee_debug_psell_build(C.c_int64(m), C.c_int64(n), colptr.ctypes.data_as(C.c_void_p), colptr.dtype.itemsize,
                                            L.ptr(rowval, L.u32p), L.ptr(nzval, L.f32p), L.ptr(ks, L.i64p) if ks is not None else None,
                                            C.byref(h)))
    v = L.PsellView()
    L.check(L.lib().polee_debug_psell_view(h, C.byref(v)))
    out = view_hashes(v)
    L.lib().polee_debug_psell_free(h)
    return out


if __name__ == "__main__":
    res = {}
    for name, smp, ks in cases():
        res[name] = build_host(smp, ks)
    json.dump(res, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    print("wrote", sys.argv[1])
