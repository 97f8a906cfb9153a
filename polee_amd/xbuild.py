"""Construction of the likelihood matrix on the GPU (polee_xbuild_*; SURVEY.md 8(f) f4, first slice): the host mirror of
the reference's `RNASeqSample(fm::FragModel, rs::Reads, ts::Transcripts, ...)` (src/rnaseq_sample.jl:390-524) for
pre-parsed inputs and the SimplisticFragModel (src/fragmodel.jl:23-169)."""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import check


class _Transcripts(C.Structure):
    _fields_ = [("n", C.c_int32), ("seq", C.c_void_p), ("strand", C.c_void_p), ("exon_ptr", C.c_void_p),
                ("exon_first", C.c_void_p), ("exon_last", C.c_void_p)]


class _Fragments(C.Structure):
    _fields_ = [("m", C.c_int64), ("seq", C.c_void_p), ("strand", C.c_void_p), ("m1_left", C.c_void_p),
                ("m1_right", C.c_void_p), ("m2_left", C.c_void_p), ("m2_right", C.c_void_p), ("m1_is_flag16", C.c_void_p),
                ("cig1_ptr", C.c_void_p), ("cig2_ptr", C.c_void_p), ("cig_op", C.c_void_p), ("cig_len", C.c_void_p)]


class _FragModel(C.Structure):
    _fields_ = [("fraglen_pmf", C.c_void_p), ("fraglen_cdf", C.c_void_p), ("fraglen_median", C.c_int32),
                ("strand_specificity", C.c_float), ("alt_frag_model", C.c_int32)]


class _BiasModel(C.Structure):
    _fields_ = [("tseq_ptr", C.c_void_p), ("tseq", C.c_void_p), ("seqbias_len", C.c_int32), ("ps_ctx", C.c_int32),
                ("orders_left", C.c_void_p), ("orders_right", C.c_void_p), ("ps_left", C.c_void_p), ("ps_right", C.c_void_p),
                ("gc_nbins", C.c_int32), ("gc_bins", C.c_void_p), ("pos_p", C.c_double), ("pos_terms", C.c_void_p),
                ("pos_maxtlen", C.c_int32), ("num_fraglens", C.c_int32), ("high_prob_fraglens", C.c_void_p),
                ("m1_reverse", C.c_void_p)]


def pack_bias(bias):
    """ctypes struct (+ the arrays it borrows) of a TRAINED bias model (polee_xb_biasmodel, include/polee_hip.h): dict with
    tseq_ptr i64 [n+1], tseq u8 (0 A, 1 C, 2 G, 3 T, 4 other; transcript orientation), orders_left / orders_right i32 [20],
    ps_left / ps_right f32 [20, 4, ps_ctx], gc_bins f32, high_prob_fraglens i32 [200], optional pos_p + pos_terms f64,
    optional m1_reverse u8 [m]."""
    keep = []

    def ptr(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)
    psl = np.ascontiguousarray(bias["ps_left"], np.float32)
    psr = np.ascontiguousarray(bias["ps_right"], np.float32)
    assert psl.ndim == 3 and psl.shape == psr.shape and psl.shape[1] == 4
    pt = bias.get("pos_terms")
    B = _BiasModel(ptr(bias["tseq_ptr"], np.int64), ptr(bias["tseq"], np.uint8), psl.shape[0], psl.shape[2],
                   ptr(bias["orders_left"], np.int32), ptr(bias["orders_right"], np.int32), ptr(psl, np.float32), ptr(psr, np.float32),
                   len(bias["gc_bins"]), ptr(bias["gc_bins"], np.float32), float(bias.get("pos_p", 0.0)), ptr(pt, np.float64),
                   0 if pt is None else len(pt), len(bias["high_prob_fraglens"]), ptr(bias["high_prob_fraglens"], np.int32),
                   ptr(bias.get("m1_reverse"), np.uint8))
    return B, keep


_T_TYPES = dict(seq=np.int32, strand=np.int8, exon_ptr=np.int64, exon_first=np.int64, exon_last=np.int64)
_F_TYPES = dict(seq=np.int32, strand=np.int8, m1_left=np.int64, m1_right=np.int64, m2_left=np.int64, m2_right=np.int64,
                m1_is_flag16=np.uint8, cig1_ptr=np.int64, cig2_ptr=np.int64, cig_op=np.uint8, cig_len=np.int32)


def pack(transcripts, fragments, fraglen_pmf, fraglen_cdf, fraglen_median, strand_specificity, alt_frag_model):
    """ctypes structs (+ the arrays they borrow) for polee_xbuild_run and for the oracle's twin of it."""
    keep = []

    def ptr(a, dt):
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)
    T = _Transcripts(int(transcripts["n"]), *[ptr(transcripts[k], dt) for k, dt in _T_TYPES.items()])
    F = _Fragments(int(fragments["m"]), *[ptr(fragments[k], dt) for k, dt in _F_TYPES.items()])
    M = _FragModel(ptr(fraglen_pmf, np.float32), ptr(fraglen_cdf, np.float32), int(fraglen_median), float(strand_specificity),
                   int(bool(alt_frag_model)))
    return T, F, M, keep


def order_mates(fragments):
    """The reference orders the mates of a pair itself -- a1 = the mate with the smaller leftpos (src/transcripts.jl:288-297)
    -- and polee_xbuild_run REQUIRES m1 to be that mate (POLEE_ERR_BAD_ARG otherwise).  A caller holding the mates in BAM
    order (mate1_idx / mate2_idx) passes through here: pairs with m2_left < m1_left are swapped, intervals and CIGAR
    ranges alike; cig2_ptr == None stands for "no second mate has operations".  Returns a new dict (arrays shared where
    nothing changes)."""
    F = dict(fragments)
    m = int(F["m"])
    c1 = np.asarray(F["cig1_ptr"], np.int64)
    c2 = np.zeros(m + 1, np.int64) if F.get("cig2_ptr") is None else np.asarray(F["cig2_ptr"], np.int64)
    F["cig2_ptr"] = c2
    m1l, m2l = np.asarray(F["m1_left"], np.int64), np.asarray(F["m2_left"], np.int64)
    swap = (m2l != 0) & (m2l < m1l)
    if not swap.any():
        return F
    for a, b in (("m1_left", "m2_left"), ("m1_right", "m2_right")):
        x, y = np.asarray(F[a], np.int64), np.asarray(F[b], np.int64)
        F[a], F[b] = np.where(swap, y, x), np.where(swap, x, y)
    l1, l2 = np.diff(c1), np.diff(c2)
    n1, n2 = np.where(swap, l2, l1), np.where(swap, l1, l2)
    s1, s2 = np.where(swap, c2[:-1], c1[:-1]), np.where(swap, c1[:-1], c2[:-1])
    p1 = np.concatenate([[0], np.cumsum(n1)])
    p2 = p1[-1] + np.concatenate([[0], np.cumsum(n2)])  # (all first mates' operations, then all second mates')
    src = np.concatenate([np.repeat(s1 - p1[:-1], n1) + np.arange(p1[-1]), np.repeat(s2 - p2[:-1], n2) + np.arange(p1[-1], p2[-1])])
    F["cig_op"] = np.asarray(F["cig_op"], np.uint8)[src] if src.size else np.zeros(1, np.uint8)
    F["cig_len"] = np.asarray(F["cig_len"], np.int32)[src] if src.size else np.zeros(1, np.int32)
    F["cig1_ptr"], F["cig2_ptr"] = p1.astype(np.int64), p2.astype(np.int64)
    return F


def build_likelihood_matrix(transcripts, fragments, fraglen_pmf, fraglen_cdf, fraglen_median, strand_specificity=0.9,
                            alt_frag_model=False, ctx=None, bias=None, return_bias=False, return_sample=False, return_tree=False):
    """-> dict(m, n, nnz, tcolptr u64 [m+1], trowval u32, tnzval f32 (the rows of X, 1-based, what RNASeqSample(xt=...)
    takes), effective_lengths f32 [n], row_fragment i64 [m], kernel_ms).  The mates of a pair may come in any order
    (order_mates).  bias: a trained bias model (pack_bias) -> the reference's default BiasedFragModel
    (src/fragmodel.jl:174-445) instead of the SimplisticFragModel; return_bias adds left_bias / right_bias (the
    transcripts' bias vectors, compute_transcript_bias!).  return_sample adds "sample": the RNASeqSample made straight from the
    result on the device (polee_loglik_create_from_xbuild: X never visits the host on its way into the likelihood); return_tree adds
    "node_parent_idxs" / "node_js": the clustering tree of the rounds variant, built from the same result on the device."""
    from .core import default_context
    ctx = ctx or default_context()
    fragments = order_mates(fragments)
    T, F, M, keep = pack(transcripts, fragments, fraglen_pmf, fraglen_cdf, fraglen_median, strand_specificity, alt_frag_model)
    h = C.c_void_p()
    if bias is not None:
        # (m1_is_flag16 and bias["m1_reverse"] describe the LONE mate of a single-end fragment -- m2_left == 0, which
        # order_mates never swaps -- so reordering the mates of pairs leaves both valid; transcripts.jl:288-297, :486)
        Bs, keep_b = pack_bias(bias)
        check(L.lib().polee_xbuild_run_biased(ctx._h, C.byref(T), C.byref(F), C.byref(M), C.byref(Bs), C.byref(h)), ctx._h)
    else:
        check(L.lib().polee_xbuild_run(ctx._h, C.byref(T), C.byref(F), C.byref(M), C.byref(h)), ctx._h)
    try:
        rows, nnz = C.c_int64(), C.c_int64()
        ms = [C.c_double(), C.c_double(), C.c_double()]
        check(L.lib().polee_xbuild_sizes(h, C.byref(rows), C.byref(nnz), *[C.byref(x) for x in ms]), ctx._h)
        out = dict(m=rows.value, n=int(transcripts["n"]), nnz=nnz.value, tcolptr=np.empty(rows.value + 1, np.uint64),
                   trowval=np.empty(nnz.value, np.uint32), tnzval=np.empty(nnz.value, np.float32),
                   effective_lengths=np.empty(int(transcripts["n"]), np.float32), row_fragment=np.empty(rows.value, np.int64),
                   kernel_ms=dict(efflen=ms[0].value, count=ms[1].value, fill=ms[2].value))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(L.lib().polee_xbuild_get(h, p(out["tcolptr"]), p(out["trowval"]), p(out["tnzval"]), p(out["effective_lengths"]),
                                       p(out["row_fragment"])), ctx._h)
        if return_tree:  # the clustering tree (rounds variant) from the result where it lies: polee_hclust_parallel_device_from_xbuild
            n_ = int(transcripts["n"])
            parents, js = np.empty(2 * n_ - 1, np.int32), np.empty(2 * n_ - 1, np.int32)
            check(L.lib().polee_hclust_parallel_device_from_xbuild(ctx._h, h, p(parents), p(js)), ctx._h)
            out["node_parent_idxs"], out["node_js"] = parents, js
        if return_sample:
            from .core import RNASeqSample
            out["sample"] = RNASeqSample(out["m"], out["n"], None, None, None, effective_lengths=out["effective_lengths"], ctx=ctx, _xbuild=h)
        if bias is not None and return_bias:
            total = int(np.asarray(bias["tseq_ptr"])[-1])
            out["left_bias"], out["right_bias"] = np.empty(total, np.float32), np.empty(total, np.float32)
            msb = C.c_double()
            check(L.lib().polee_xbuild_get_bias(h, p(out["left_bias"]), p(out["right_bias"]), C.byref(msb)), ctx._h)
            out["kernel_ms"]["bias"] = msb.value
    finally:
        L.lib().polee_xbuild_destroy(h)
    del keep
    return out
