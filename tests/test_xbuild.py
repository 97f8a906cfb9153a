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
    g = XB.build_likelihood_matrix(d["transcripts"], d["fragments"], pmf, cdf, med, 0.85, alt, ctx=ctx, return_sample=True, return_tree=True)
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
    # (... without visiting the host: polee_loglik_create_from_xbuild lays the result out where xbuild left it)
    sd = g["sample"]
    assert sd.built_on_device and sd.info["nnz"] == g["nnz"]
    # ... and so does the tree: the rounds variant from the result on the device = from its columns on the host
    ph, jh = P.hclust(g["m"], n, (X.indptr + 1).astype(np.uint64), (X.indices + 1).astype(np.uint32), parallel=True)
    np.testing.assert_array_equal(g["node_parent_idxs"], ph)
    np.testing.assert_array_equal(g["node_js"], jh)
    lpd, gradd = sd.log_likelihood(x)
    for k in range(2):
        lpo, go = so.log_likelihood(x[k])
        assert abs(lp[k] - lpo) <= 1e-6 * abs(lpo) and abs(lpd[k] - lpo) <= 1e-6 * abs(lpo)
        np.testing.assert_allclose(grad[k], go, rtol=1e-4, atol=1e-6 * np.abs(go).max())
        np.testing.assert_allclose(gradd[k], go, rtol=1e-4, atol=1e-6 * np.abs(go).max())


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


# ---- BiasedFragModel (the reference's default; src/fragmodel.jl:174-445 with a trained bias model, src/bias.jl) ------------

def _toy_bias(T, F, pmf, **kw):
    b = synth_aln.make_bias_model(T, F, pmf, **kw)
    return b


def test_oracle_biased_model_on_a_hand_computed_case():
    """One transcript on the + strand (exons 101..160 and 201..260: 120 bases), a trained model with ONE position in each
    sequence-bias table, two GC bins, no positional model; a paired fragment.  Every number is recomputed in NumPy from
    the reference's formulas: compute_transcript_bias! (bias.jl:834-858), effective_length (fragmodel.jl:372-410),
    genomic_to_transcriptomic (transcripts.jl:452-538) and condfragprob (fragmodel.jl:413-445)."""
    T = _one_transcript([(101, 160), (201, 260)])
    M_, N_ = 0, 3
    F = _frags([(1, (111, 150, [(M_, 40)]), (211, 250, [(M_, 40)]), 0)])   # fragment 111..250 minus the 40-base intron: 100 bases
    pmf, cdf, med = synth_aln.fraglen_model(100.0, 10.0)
    rng = np.random.default_rng(0)
    tseq = rng.integers(0, 4, 120).astype(np.uint8)
    orders_l = np.full(20, -1, np.int32); orders_l[5] = 0      # the fragment's first base itself (pos - 5 + 5)
    orders_r = np.full(20, -1, np.int32); orders_r[14] = 1     # the fragment's last base with one base of context
    ps_l = np.ones((20, 4, 4), np.float32); ps_l[5, :, 0] = [0.5, 1.0, 1.5, 2.0]
    ps_r = np.ones((20, 4, 4), np.float32); ps_r[14] = rng.uniform(0.5, 2.0, (4, 4)).astype(np.float32)
    bias = dict(tseq_ptr=np.array([0, 120], np.int64), tseq=tseq, orders_left=orders_l, orders_right=orders_r, ps_left=ps_l,
                ps_right=ps_r, gc_bins=np.array([0.8, 1.25], np.float32),
                high_prob_fraglens=np.array([100, 99, 101, 130], np.int32), m1_reverse=np.zeros(1, np.uint8))
    Ts, Fs, Ms, keep = XB.pack(T, F, pmf, cdf, med, 0.9, False)
    Bs, kb = XB.pack_bias(bias)
    o = OX.build_biased(Ts, Fs, Ms, Bs, 1, 120)
    code = lambda j: int(tseq[j - 1]) if 1 <= j <= 120 else 0          # off the ends: A
    left = np.array([ps_l[5, code(p), 0] for p in range(1, 121)], np.float32)
    right = np.array([ps_r[14, code(p), code(p + 1)] for p in range(1, 121)], np.float32)
    np.testing.assert_array_equal(o["left_bias"], left)
    np.testing.assert_array_equal(o["right_bias"], right)
    gcb = lambda x: bias["gc_bins"][int(np.clip(np.round(x * 2), 1, 2)) - 1]  # round half to even, like Julia's round(Int, .)
    isgc = (tseq == 1) | (tseq == 2)
    el = np.float32(0)
    for fl in (100, 99, 101):                                             # 130 > tlen: skipped
        c = np.float32(0)
        for pos in range(1, 120 - fl + 2):
            prop = np.float32(isgc[pos - 1:pos - 1 + fl].sum() / fl)        # (the oracle slides the window in Float32: <= 1e-5 apart)
            c = np.float32(c + np.float32(np.float32(left[pos - 1] * right[pos + fl - 2]) * gcb(prop)))
        el = np.float32(el + np.float32(c * pmf[fl - 1]))
    np.testing.assert_allclose(o["effective_lengths"][0], el, rtol=1e-5)
    # the fragment: positions 11 .. 110 of the transcript (111 - 101 + 1 = 11; 100 bases)
    gc = isgc[10:110].mean()
    expect = np.float32(0.9) * pmf[99] * np.float32(left[10] * right[109]) * gcb(gc) / o["effective_lengths"][0]
    assert o["m"] == 1 and o["trowval"].tolist() == [1]
    np.testing.assert_allclose(o["tnzval"][0], expect, rtol=1e-6)


def test_oracle_biased_single_end_and_negative_strand():
    """genomic_to_transcriptomic for a transcript on the - strand and single-end reads (transcripts.jl:471-512): the position
    arithmetic as written, the overhang nudges; all entries finite, probabilities scale with 1 / effective length."""
    d = synth_aln.make(40, 1500, seed=8, p_single=0.5)
    pmf, cdf, med = synth_aln.fraglen_model()
    bias = _toy_bias(d["transcripts"], d["fragments"], pmf, seed=4, use_pos_bias=True)
    Ts, Fs, Ms, keep = XB.pack(d["transcripts"], d["fragments"], pmf, cdf, med, 0.9, False)
    Bs, kb = XB.pack_bias(bias)
    o = OX.build_biased(Ts, Fs, Ms, Bs, 40, int(bias["tseq_ptr"][-1]))
    assert np.isfinite(o["tnzval"]).all() and (o["tnzval"] > 1e-12).all() and (o["effective_lengths"] >= 1).all()
    assert (np.asarray(d["transcripts"]["strand"]) < 0).any() and (np.asarray(d["fragments"]["m2_left"]) == 0).sum() > 300
    tt = d["true_transcript"]
    ptr = o["tcolptr"].astype(np.int64) - 1
    hit = [tt[i] + 1 in o["trowval"][ptr[r]:ptr[r + 1]] for r, i in enumerate(o["row_fragment"]) if tt[i] >= 0]
    assert np.mean(hit) > 0.98  # (a handful of single-end guesses fall off the transcript's end and are dropped)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,pos,seed", [(200, 15000, False, 3), (1500, 80000, True, 4)])
def test_device_biased_model_matches_oracle(n, m, pos, seed):
    """polee_xbuild_run_biased against the restatement: bit-identical bias vectors and effective lengths (the same Float32
    operations in the same order), identical sparsity pattern, values to 1e-6."""
    import polee_amd as P
    ctx = P.Context(0)
    d = synth_aln.make(n, m, seed=seed, p_single=0.2, p_noise=0.05)
    pmf, cdf, med = synth_aln.fraglen_model(180.0, 60.0)
    bias = _toy_bias(d["transcripts"], d["fragments"], pmf, seed=seed + 10, use_pos_bias=pos)
    Ts, Fs, Ms, keep = XB.pack(d["transcripts"], d["fragments"], pmf, cdf, med, 0.85, False)
    Bs, kb = XB.pack_bias(bias)
    total = int(bias["tseq_ptr"][-1])
    o = OX.build_biased(Ts, Fs, Ms, Bs, n, total)
    g = XB.build_likelihood_matrix(d["transcripts"], d["fragments"], pmf, cdf, med, 0.85, False, ctx=ctx, bias=bias, return_bias=True)
    np.testing.assert_array_equal(g["right_bias"], o["right_bias"])
    if pos:  # (the positional term is Float64 arithmetic with a pow(): device and host libm may differ in the last place)
        np.testing.assert_allclose(g["left_bias"], o["left_bias"], rtol=2e-7)
        np.testing.assert_allclose(g["effective_lengths"], o["effective_lengths"], rtol=1e-5)
    else:
        np.testing.assert_array_equal(g["left_bias"], o["left_bias"])
        np.testing.assert_array_equal(g["effective_lengths"], o["effective_lengths"])
    assert g["m"] == o["m"] and g["nnz"] == o["nnz"] and g["m"] > 0.8 * m
    np.testing.assert_array_equal(g["row_fragment"], o["row_fragment"])
    np.testing.assert_array_equal(g["trowval"], o["trowval"])
    np.testing.assert_allclose(g["tnzval"], o["tnzval"], rtol=1e-6 if not pos else 2e-5)
    print("biased X construction n=%d m=%d: %d rows, %d non-zeros, %d bases; kernels %s ms" % (n, m, g["m"], g["nnz"], total, g["kernel_ms"]))
    # a malformed model is rejected
    bad = dict(bias); bad["tseq_ptr"] = bias["tseq_ptr"].copy(); bad["tseq_ptr"][-1] += 5
    bad["tseq"] = np.concatenate([bias["tseq"], np.zeros(5, np.uint8)])
    with pytest.raises(P.PoleeError):
        XB.build_likelihood_matrix(d["transcripts"], d["fragments"], pmf, cdf, med, 0.85, False, ctx=ctx, bias=bad)
