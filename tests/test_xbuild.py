"""X construction (first slice).  CPU: the oracle restatement (oracle/xbuild_oracle.c) on hand-built
cases with known answers -- fragment length through a spliced alignment, intron encroachment, single-end reads, strand,
effective length.  GPU (-m gpu): the HIP path (polee_xbuild_run) against the oracle on synthetic alignments -- identical
sparsity pattern, values to 1e-6 -- and the matrix it builds fed to the likelihood."""
import numpy as np
import pytest

from oracle import xbuild as OX
from polee_amd import xbuild as XB
from tools import synth_aln


def _one_transcript(exons, strand=1):
    ef, el = zip(*exons)
    return dict(n=1, seq=[0], strand=[strand], exon_ptr=[0, len(exons)], exon_first=list(ef), exon_last=list(el))


def _frags(items):
    """items: (strand, m1 (left, right, [(op, len), ...]), m2 or None, flag16)"""
    F = dict(m=len(items), seq=[0] * len(items), strand=[], m1_left=[], m1_right=[], m2_left=[], m2_right=[], m1_is_flag16=[],
             cig1_ptr=[0], cig2_ptr=[], cig_op=[], cig_len=[])
    ops2 = []
    for strand, m1, m2, f16 in items:
        F["strand"].append(strand); F["m1_left"].append(m1[0]); F["m1_right"].append(m1[1]); F["m1_is_flag16"].append(f16)
        F["cig_op"] += [o for o, _ in m1[2]]; F["cig_len"] += [l for _, l in m1[2]]
        F["cig1_ptr"].append(len(F["cig_op"]))
        F["m2_left"].append(m2[0] if m2 else 0); F["m2_right"].append(m2[1] if m2 else 0)
        ops2.append(m2[2] if m2 else [])
    F["cig2_ptr"] = [len(F["cig_op"])]
    for o in ops2:
        F["cig_op"] += [a for a, _ in o]; F["cig_len"] += [b for _, b in o]
        F["cig2_ptr"].append(len(F["cig_op"]))
    if not F["cig_op"]:
        F["cig_op"], F["cig_len"] = [0], [0]
    return F


def _run_oracle(T, F, ss=0.9, alt=False):
    pmf, cdf, med = synth_aln.fraglen_model()
    Ts, Fs, Ms, keep = XB.pack(T, F, pmf, cdf, med, ss, alt)
    return OX.build(Ts, Fs, Ms, T["n"]), pmf, cdf, med


def test_oracle_effective_length_and_spliced_fragment():
    T = _one_transcript([(100, 199), (300, 399), (1000, 1299)])
    M, N = 0, 3
    F = _frags([
        (1, (150, 324, [(M, 50), (N, 100), (M, 25)]), (350, 399, []), 0),          # spliced pair: fragment 150..399 minus the intron
        (1, (150, 224, [(M, 75)]), None, 0),                                         # runs into the intron: incompatible
        (1, (150, 201, [(M, 52)]), (1000, 1049, []), 0),                             # overhang of 2 bases: allowed (:275)
        (1, (150, 202, [(M, 53)]), (1000, 1049, []), 0),                             # 3 bases: not
        (-1, (320, 394, []), None, 1),                                               # single-end, flag 16, other strand
        (1, (320, 394, []), None, 0),                                                # single-end, forward
        (1, (90, 164, []), (300, 350, []), 0),                                       # starts before the transcript
    ])
    o, pmf, cdf, med = _run_oracle(T, F)
    tlen = 100 + 100 + 300
    el = np.float32(0)
    for l in range(1, tlen + 1):
        el = np.float32(el + np.float32(pmf[l - 1] * np.float32(tlen - l + 1)))
    assert o["effective_lengths"][0] == el
    rows = {int(i): r for r, i in enumerate(o["row_fragment"])}
    assert sorted(rows) == [0, 2, 4, 5], rows
    val = lambda i: float(o["tnzval"][int(o["tcolptr"][rows[i]]) - 1])
    np.testing.assert_allclose(val(0), 0.9 * pmf[150 - 1] / el, rtol=1e-6)           # 399 - 150 + 1 - 100 intronic
    np.testing.assert_allclose(val(2), 0.9 * pmf[(1049 - 150 + 1) - (100 + 600) - 1] / el, rtol=1e-6)
    np.testing.assert_allclose(val(4), (1.0 - np.float32(0.9)) * pmf[min(394 - 100 + 1, med) - 1] / el, rtol=1e-6)
    np.testing.assert_allclose(val(5), 0.9 * pmf[min(1299 - 320 + 1, med) - 1] / el, rtol=1e-6)


def test_oracle_soft_clips_and_alt_model():
    T = _one_transcript([(100, 399)])
    M, S = 0, 4
    F = _frags([(1, (150, 224, [(S, 5), (M, 70)]), (300, 374, [(M, 70), (S, 5)]), 0),   # clips at both outer ends
                (1, (150, 224, [(M, 30), (3, 10), (M, 35)]), (300, 374, []), 0)])       # an N inside an exon: incompatible
    o, pmf, cdf, med = _run_oracle(T, F, alt=True)
    assert o["row_fragment"].tolist() == [0]
    tlen = 300
    el = np.float32(0)
    for l in range(1, tlen + 1):
        el = np.float32(el + np.float32(np.float32(pmf[l - 1] / cdf[tlen - 1]) * np.float32(tlen - l + 1)))
    assert o["effective_lengths"][0] == max(el, np.float32(1))
    np.testing.assert_allclose(o["tnzval"][0], 0.9 * pmf[(374 - 150 + 1) - 1] / el / cdf[tlen - 1], rtol=1e-6)


def test_oracle_recovers_the_source_transcript_of_synthetic_fragments():
    d = synth_aln.make(200, 5000, seed=5)
    pmf, cdf, med = synth_aln.fraglen_model()
    Ts, Fs, Ms, keep = XB.pack(d["transcripts"], d["fragments"], pmf, cdf, med, 0.9, False)
    o = OX.build(Ts, Fs, Ms, 200)
    ptr = o["tcolptr"].astype(np.int64) - 1
    tt = d["true_transcript"]
    assert (tt[o["row_fragment"]] >= 0).mean() > 0.99
    for r, i in enumerate(o["row_fragment"]):
        if tt[i] >= 0:
            assert tt[i] + 1 in o["trowval"][ptr[r]:ptr[r + 1]]
    assert set(np.flatnonzero(tt >= 0)) <= set(o["row_fragment"].tolist())
    assert (np.diff(o["tcolptr"].astype(np.int64)) >= 1).all()  # no empty rows (compact_indexes!)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,alt,seed", [(300, 20000, False, 3), (2500, 150000, True, 4)])
def test_device_matches_oracle(n, m, alt, seed):
    import polee_amd as P
    ctx = P.Context(0)
    d = synth_aln.make(n, m, seed=seed, p_single=0.15, p_noise=0.08)
    pmf, cdf, med = synth_aln.fraglen_model(180.0, 60.0)
    Ts, Fs, Ms, keep = XB.pack(d["transcripts"], d["fragments"], pmf, cdf, med, 0.85, alt)
    o = OX.build(Ts, Fs, Ms, n)
    g = XB.build_likelihood_matrix(d["transcripts"], d["fragments"], pmf, cdf, med, 0.85, alt, ctx=ctx)
    assert g["m"] == o["m"] and g["nnz"] == o["nnz"] and g["m"] > 0.8 * m
    np.testing.assert_array_equal(g["row_fragment"], o["row_fragment"])
    np.testing.assert_array_equal(g["tcolptr"], o["tcolptr"])
    np.testing.assert_array_equal(g["trowval"], o["trowval"])          # identical sparsity pattern
    np.testing.assert_array_equal(g["effective_lengths"], o["effective_lengths"])  # the same Float32 sums
    np.testing.assert_allclose(g["tnzval"], o["tnzval"], rtol=1e-6)
    print("X construction n=%d m=%d: %d rows, %d non-zeros; kernels %s ms" % (n, m, g["m"], g["nnz"], g["kernel_ms"]))
    # ... and the matrix goes straight into the likelihood
    from oracle import oracle as O
    import scipy.sparse as sp
    s = P.RNASeqSample(g["m"], n, None, None, None, g["effective_lengths"], ctx=ctx, xt=(g["tcolptr"], g["trowval"], g["tnzval"]))
    X = sp.csr_matrix((g["tnzval"], g["trowval"].astype(np.int64) - 1, g["tcolptr"].astype(np.int64) - 1), shape=(g["m"], n)).tocsc()
    X.sort_indices()
    so = O.Sample(g["m"], n, (X.indptr + 1).astype(np.uint64), (X.indices + 1).astype(np.uint32), X.data.astype(np.float32))
    x = np.clip(np.random.default_rng(0).dirichlet(np.ones(n), size=2), 1e-10, 1).astype(np.float32)
    lp, grad = s.log_likelihood(x)
    for k in range(2):
        lpo, go = so.log_likelihood(x[k])
        assert abs(lp[k] - lpo) <= 1e-6 * abs(lpo)
        np.testing.assert_allclose(grad[k], go, rtol=1e-4, atol=1e-6 * np.abs(go).max())


def _bam_order(F, rng):
    """The same fragments with the mates of half of the pairs exchanged (what a caller holding mate1 / mate2 in BAM order
    has): intervals and CIGAR ranges swapped, offsets rebuilt (first mates' operations, then second mates')."""
    m = F["m"]
    c1, c2 = np.asarray(F["cig1_ptr"]), np.asarray(F["cig2_ptr"])
    sw = (np.asarray(F["m2_left"]) != 0) & (rng.random(m) < 0.5)
    G = dict(F)
    for a, b in (("m1_left", "m2_left"), ("m1_right", "m2_right")):
        x, y = np.asarray(F[a]), np.asarray(F[b])
        G[a], G[b] = np.where(sw, y, x), np.where(sw, x, y)
    ops1, ops2 = [], []
    for i in range(m):
        r1, r2 = (c1[i], c1[i + 1]), (c2[i], c2[i + 1])
        if sw[i]:
            r1, r2 = r2, r1
        ops1.append(r1); ops2.append(r2)
    src, p1, p2 = [], [0], []
    for a, b in ops1:
        src += range(a, b); p1.append(len(src))
    p2.append(len(src))
    for a, b in ops2:
        src += range(a, b); p2.append(len(src))
    src = np.array(src, np.int64)
    G["cig_op"] = np.asarray(F["cig_op"])[src] if src.size else np.zeros(1, np.uint8)
    G["cig_len"] = np.asarray(F["cig_len"])[src] if src.size else np.zeros(1, np.int32)
    G["cig1_ptr"], G["cig2_ptr"] = np.array(p1, np.int64), np.array(p2, np.int64)
    return G, sw


def test_order_mates_restores_the_leftmost_first_contract():
    """the reference orders a pair's mates by leftpos itself (transcripts.jl:288-297); the C ABI requires that
    order and the host wrapper establishes it.  Mates shuffled into 'BAM order' and passed through order_mates give the
    oracle the same matrix as the original fragments."""
    d = synth_aln.make(150, 4000, seed=9, p_single=0.2)
    pmf, cdf, med = synth_aln.fraglen_model()
    F = d["fragments"]
    G, sw = _bam_order(F, np.random.default_rng(2))
    assert sw.sum() > 1000 and (np.asarray(G["m2_left"])[sw] < np.asarray(G["m1_left"])[sw]).any()
    H = XB.order_mates(G)
    lft = np.asarray(H["m2_left"])
    assert ((lft == 0) | (lft >= np.asarray(H["m1_left"]))).all()
    outs = []
    for frag in (F, H):
        Ts, Fs, Ms, keep = XB.pack(d["transcripts"], frag, pmf, cdf, med, 0.9, False)
        outs.append(OX.build(Ts, Fs, Ms, 150))
    for k in ("tcolptr", "trowval", "tnzval", "row_fragment"):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
    # all single-end, cig2_ptr omitted
    S = dict(F, m2_left=np.zeros(F["m"], np.int64), m2_right=np.zeros(F["m"], np.int64), cig2_ptr=None)
    assert (XB.order_mates(S)["cig2_ptr"] == 0).all()


@pytest.mark.gpu
def test_device_validates_fragments_and_accepts_any_mate_order():
    import polee_amd as P
    from polee_amd import _lib as L
    import ctypes as C
    ctx = P.Context(0)
    d = synth_aln.make(300, 20000, seed=3, p_single=0.15)
    pmf, cdf, med = synth_aln.fraglen_model(180.0, 60.0)
    ref = XB.build_likelihood_matrix(d["transcripts"], d["fragments"], pmf, cdf, med, 0.85, False, ctx=ctx)
    G, sw = _bam_order(d["fragments"], np.random.default_rng(5))
    got = XB.build_likelihood_matrix(d["transcripts"], G, pmf, cdf, med, 0.85, False, ctx=ctx)  # (wrapper orders the mates)
    for k in ("tcolptr", "trowval", "tnzval", "row_fragment"):
        np.testing.assert_array_equal(ref[k], got[k])

    def raw(F):
        Ts, Fs, Ms, keep = XB.pack(d["transcripts"], F, pmf, cdf, med, 0.85, False)
        h = C.c_void_p()
        rc = L.lib().polee_xbuild_run(ctx._h, C.byref(Ts), C.byref(Fs), C.byref(Ms), C.byref(h))
        if rc == 0:
            L.lib().polee_xbuild_destroy(h)
        return rc
    assert raw(d["fragments"]) == 0
    assert raw(G) != 0  # mates not ordered: rejected, not silently a different X
    bad = dict(d["fragments"]); c = np.array(bad["cig1_ptr"]); c[5] = c[4] - 1; bad["cig1_ptr"] = c
    assert raw(bad) != 0  # offsets not monotone
    bad = dict(d["fragments"]); r = np.array(bad["m1_right"]); r[7] = np.asarray(bad["m1_left"])[7] - 1; bad["m1_right"] = r
    assert raw(bad) != 0  # right < left
    bad = dict(d["fragments"]); ln = np.array(bad["cig_len"]); ln[0] = -3; bad["cig_len"] = ln
    assert raw(bad) != 0 or np.diff(np.asarray(bad["cig1_ptr"]))[0] == 0  # negative operation length
