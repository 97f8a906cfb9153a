"""Synthetic alignments for the X-construction path (tools/synth_aln.c): bench / test support, not part of the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libpolee_synth_aln.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_LIB)
        _lib.aln_model_create.restype = C.c_void_p
        _lib.aln_model_num_exons.restype = C.c_int64
        _lib.aln_fragments.restype = C.c_int64
    return _lib


def fraglen_model(mean=200.0, sd=40.0):
    """Fragment length pmf / cdf / median as SimplisticFragModel holds them (src/fragmodel.jl:23-115): f32 [2000]."""
    l = np.arange(1, 2001, dtype=np.float64)
    pmf = np.exp(-0.5 * ((l - mean) / sd) ** 2)
    pmf[:20] = 0
    pmf = (pmf / pmf.sum()).astype(np.float32)
    cdf = pmf.copy()
    for i in range(1, 2000):  # (Float32 accumulation as fragmodel.jl:106-108)
        cdf[i] = np.float32(cdf[i] + cdf[i - 1])
    median = int(np.searchsorted(cdf, 0.5, side="left")) + 1
    return pmf, cdf, median


def make(n, m, num_seq=4, seed=1, read_len=75, p_single=0.1, p_noise=0.05, strand_specificity=0.9, pmf=None):
    """dict(transcripts=..., fragments=..., true_transcript i32 [m]) in the layout of polee_xb_transcripts / _fragments."""
    L = lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    h = C.c_void_p(L.aln_model_create(C.c_int32(n), C.c_int32(num_seq), C.c_uint64(seed)))
    nex = L.aln_model_num_exons(h)
    T = dict(n=n, seq=np.empty(n, np.int32), strand=np.empty(n, np.int8), exon_ptr=np.empty(n + 1, np.int64),
             exon_first=np.empty(nex, np.int64), exon_last=np.empty(nex, np.int64))
    L.aln_model_get(h, p(T["seq"]), p(T["strand"]), p(T["exon_ptr"]), p(T["exon_first"]), p(T["exon_last"]))
    if pmf is None:
        pmf = fraglen_model()[0]
    F = dict(m=m, seq=np.empty(m, np.int32), strand=np.empty(m, np.int8), m1_left=np.empty(m, np.int64),
             m1_right=np.empty(m, np.int64), m2_left=np.empty(m, np.int64), m2_right=np.empty(m, np.int64),
             m1_is_flag16=np.empty(m, np.uint8), cig1_ptr=np.empty(m + 1, np.int64), cig2_ptr=np.empty(m + 1, np.int64))
    cop = np.empty(24 * m + 8, np.uint8)
    clen = np.empty(24 * m + 8, np.int32)
    tt = np.empty(m, np.int32)
    nc = L.aln_fragments(h, C.c_int64(m), C.c_uint64(seed), p(np.ascontiguousarray(pmf, np.float32)), C.c_int(read_len),
                         C.c_double(p_single), C.c_double(p_noise), C.c_double(strand_specificity), p(F["seq"]), p(F["strand"]),
                         p(F["m1_left"]), p(F["m1_right"]), p(F["m2_left"]), p(F["m2_right"]), p(F["m1_is_flag16"]),
                         p(F["cig1_ptr"]), p(F["cig2_ptr"]), p(cop), p(clen), p(tt))
    F["cig_op"], F["cig_len"] = cop[:max(nc, 1)].copy(), clen[:max(nc, 1)].copy()
    L.aln_model_free(h)
    return dict(transcripts=T, fragments=F, true_transcript=tt)


def make_bias_model(transcripts, fragments, pmf, seed=1, use_pos_bias=False, num_fraglens=200, max_order=2, gc_nbins=20):
    """A synthetic TRAINED bias model in the layout of polee_xb_biasmodel (include/polee_hip.h): random transcript sequences
    (codes 0..3, a few N = 4), sequence-bias tables of the reference's shape (20 positions, orders in -1..max_order,
    probability ratios in [0.8, 1.25]), GC histogram bins, optionally a positional model, the `num_fraglens` most probable
    fragment lengths (fragmodel.jl:358-359), and m1_reverse for the single-end fragments."""
    rng = np.random.default_rng(seed)
    ep, ef, el = (np.asarray(transcripts[k], np.int64) for k in ("exon_ptr", "exon_first", "exon_last"))
    n = int(transcripts["n"])
    lens = np.array([int((el[ep[j]:ep[j + 1]] - ef[ep[j]:ep[j + 1]] + 1).sum()) for j in range(n)], np.int64)
    tseq_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gc_rate = rng.uniform(0.35, 0.65, n)
    tseq = np.empty(int(tseq_ptr[-1]), np.uint8)
    for j in range(n):
        g = rng.random(lens[j]) < gc_rate[j]
        tseq[tseq_ptr[j]:tseq_ptr[j + 1]] = np.where(g, rng.integers(1, 3, lens[j]), rng.integers(0, 2, lens[j]) * 3)
    tseq[rng.random(tseq.size) < 0.001] = 4
    L, ctx = 20, 4 ** max_order
    orders = lambda: rng.choice(np.arange(-1, max_order + 1), L).astype(np.int32)
    ps = lambda: rng.uniform(0.8, 1.25, (L, 4, ctx)).astype(np.float32)
    out = dict(tseq_ptr=tseq_ptr, tseq=tseq, orders_left=orders(), orders_right=orders(), ps_left=ps(), ps_right=ps(),
               gc_bins=rng.uniform(0.5, 1.5, gc_nbins).astype(np.float32),
               high_prob_fraglens=(np.argsort(-np.asarray(pmf), kind="stable")[:num_fraglens] + 1).astype(np.int32),
               m1_reverse=(rng.random(int(fragments["m"])) < 0.5).astype(np.uint8))
    if use_pos_bias:
        p = 2e-4
        maxt = int(lens.max())
        out["pos_p"] = p
        out["pos_terms"] = np.cumsum(p * (1 - p) ** np.arange(maxt) / np.arange(1, maxt + 1)).astype(np.float64)
    return out
