"""The device-side layout builder (polee_amd/csrc/psell_device.hip) against the host builder (psell_build.cpp), which is its
checker: every mix of host and device stages must give the same bytes (polee_debug_psell_build_device), at small sizes for
every kind of input the builder distinguishes, with several segments per stream, and at BASELINE's C2 size.  The product
path (polee_loglik_create & co.) builds on the device by default: the parity tests against the oracle run through it."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import polee_amd
    return polee_amd


@pytest.fixture(scope="module")
def ctx(P):
    return P.Context(0)


@pytest.fixture(scope="module")
def tools():
    from tools.probe import device_build_check as D
    from tools.probe import layout_hash as H
    return D, H


@pytest.fixture(scope="module")
def cases(tools):
    return {name: (smp, ks) for name, smp, ks in tools[1].cases()}


def _same(a, b):
    return [k for k in a if a[k] != b[k] and not (k == "single_logsum" and abs(a[k] - b[k]) <= 1e-12 * max(1.0, abs(a[k])))]


@pytest.mark.parametrize("mask", [1, 2, 4, 7])
def test_every_mix_of_host_and_device_stages_gives_the_host_builders_bytes(P, ctx, tools, cases, mask):
    """bit 0: keys / stable sort / runs; bit 1: greedy packing of leftover rows; bit 2: tiles, dictionaries, emission.  Inputs:
    the generator's patterns, every fragment its own subset, per-entry dropout, multiplicities (factored likelihood), wide sets,
    rows without structure (the host builder's case: the device builder hands over), long rows, the real fixture tiled.
    (single_logsum, the one float the builders SUM, to 1e-12: the device's log is not libm's.)"""
    D, H = tools
    assert len(cases) >= 8
    for name, (smp, ks) in cases.items():
        host, _ = D.build(ctx, smp, ks, -1)
        dev, _ = D.build(ctx, smp, ks, mask)
        assert _same(host, dev) == [], (name, mask, _same(host, dev))


def test_several_segments_per_stream(P):
    """POLEE_PSELL_SEG_ROWS (read once per process by both builders): 20 000 rows per segment instead of 262 144 -- tiles that
    close at segment ends, dictionaries restarted, the per-segment scratch of the device builder."""
    env = dict(os.environ, POLEE_PSELL_SEG_ROWS="20000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "device_build_check.py"), "7"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "mismatching cases: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_c2_size_layouts_are_identical(P, ctx, tools):
    """BASELINE's C2 (20 M fragments x 200 k transcripts): the generator as built and every fragment its own subset (all rows
    through the packing), all stages on the device."""
    D, H = tools
    from tools import synth
    for literal in (False, True):
        smp = synth.make_sample(200000, 20000000, 8.0, 123456789, literal=literal)
        host, _ = D.build(ctx, smp, None, -1)
        dev, _ = D.build(ctx, smp, None, 7)
        assert _same(host, dev) == [], (literal, _same(host, dev))


def test_product_path_builds_on_the_device_and_hands_unstructured_matrices_to_the_host(P, ctx, cases):
    from tools import synth
    smp, _ = cases["patterns"]
    colptr, rowval, nzval = synth.to_csc(smp)
    s = P.RNASeqSample(smp["m"], smp["n"], colptr, rowval, nzval, ctx=ctx)
    assert s.built_on_device
    sx = P.RNASeqSample(smp["m"], smp["n"], None, None, None, ctx=ctx, xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    assert sx.built_on_device
    x = np.random.default_rng(3).dirichlet(np.ones(smp["n"])).astype(np.float32)
    lp, g = s.log_likelihood(x)
    lpx, gx = sx.log_likelihood(x)
    assert abs(lp - lpx) <= 1e-6 * abs(lp) and np.allclose(g, gx, rtol=1e-4, atol=1e-3 * np.abs(g).max())
    smp, _ = cases["random"]
    colptr, rowval, nzval = synth.to_csc(smp)
    s = P.RNASeqSample(smp["m"], smp["n"], colptr, rowval, nzval, ctx=ctx)
    assert not s.built_on_device and s.info["stream_rows"][6] == smp["m"]  # (stream C: kept in CSR)


def test_errors_of_the_device_path_are_the_host_builders(P, ctx):
    m, n = 4, 5
    colptr = np.array([1, 3, 5, 5, 6, 7], np.uint64)
    rowval = np.array([1, 2, 2, 3, 9, 4], np.uint32)  # row 9 of 4
    nzval = np.ones(6, np.float32)
    with pytest.raises(Exception, match="rowval out of range"):
        P.RNASeqSample(m, n, colptr, rowval, nzval, ctx=ctx)
    tcolptr = np.array([1, 3, 4], np.uint64)
    with pytest.raises(Exception, match="strictly ascending"):
        P.RNASeqSample(2, 5, None, None, None, ctx=ctx, xt=(tcolptr, np.array([3, 2, 1], np.uint32), np.ones(3, np.float32)))
    with pytest.raises(Exception, match="out of range"):
        P.RNASeqSample(2, 5, None, None, None, ctx=ctx, xt=(tcolptr, np.array([2, 9, 1], np.uint32), np.ones(3, np.float32)))
    # an empty matrix and a matrix of empty rows build (nothing to lay out)
    s = P.RNASeqSample(3, 4, np.ones(5, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.float32), ctx=ctx)
    assert s.info["num_slices"] == 0


def test_device_buffers_are_kept_between_samples_and_freed_on_request(P, ctx, cases):
    """common.hpp, DevBlockCache: what a build frees on the device is kept by size class (no hipFree / hipMalloc per buffer, whose
    multi-second stalls every few samples DESIGN 5.1 records); a second build of the same sample allocates nothing new; the trim
    hands everything back; results do not depend on whether blocks are fresh or reused."""
    from polee_amd import core
    from tools import synth
    smp, _ = cases["literal"]
    colptr, rowval, nzval = synth.to_csc(smp)
    core.host_cache_trim()
    assert core.device_cache_bytes() == 0
    x = np.random.default_rng(5).dirichlet(np.ones(smp["n"])).astype(np.float32)
    s1 = P.RNASeqSample(smp["m"], smp["n"], colptr, rowval, nzval, ctx=ctx)
    s1.set_deterministic(True)
    lp1, g1 = s1.log_likelihood(x)
    t1 = P.hclust(smp["m"], smp["n"], colptr, rowval, device=True, ctx=ctx)
    kept1 = core.device_cache_bytes()
    assert kept1 > 0
    del s1
    s2 = P.RNASeqSample(smp["m"], smp["n"], colptr, rowval, nzval, ctx=ctx)  # (every block it needs is in the cache, dirty)
    s2.set_deterministic(True)
    lp2, g2 = s2.log_likelihood(x)
    t2 = P.hclust(smp["m"], smp["n"], colptr, rowval, device=True, ctx=ctx)
    assert lp1 == lp2 and np.array_equal(g1, g2)
    assert np.array_equal(t1[0], t2[0]) and np.array_equal(t1[1], t2[1])
    del s2
    kept2 = core.device_cache_bytes()
    assert kept2 <= 1.3 * (kept1 + 2 ** 26), (kept1, kept2)  # no growth from sample to sample
    core.host_cache_trim()
    assert core.device_cache_bytes() == 0


def test_random_matrices_through_both_device_builders(P):
    """tools/probe/fuzz_device_builders.py: random matrices of many shapes (unstructured; genes of up to 4 / 12 / 20 / 40 / 70 isoforms
    with patterns, random subsets, strays and empty fragments; tiny and empty matrices; multiplicities; fragment order shuffled):
    layouts byte for byte, trees node for node, against the host builders."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "fuzz_device_builders.py"), "60", "7"], capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0 and "mismatching 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_shared_x_with_the_device_layout_builder_switched_off():
    """with POLEE_DEVICE_BUILD=0 (documented in README / cohort.py) and the tree on the device, sample_and_tree sends the
    layout through polee_loglik_create_from_devx -- which must then build it on the host from the handle's arrays instead of
    failing: the same fit as with the device builder (a subprocess: the switch is read once per process)."""
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
import polee_amd as P
from tools import synth
s = synth.make_sample(3000, 150000, 8.0, 5)
c, r, v = synth.to_csc(s)
ctx = P.Context(0)
smp, tree = P.sample_and_tree(P.LogitSkewNormalPTTApprox("cluster_device"), s["m"], s["n"], c, r, v, s["effective_lengths"], ctx=ctx)
xs = np.full((2, s["n"]), 1.0 / s["n"], np.float32)
lp, g = smp.log_likelihood(xs)
print("RESULT %%d %%r %%r" %% (int(smp.built_on_device), float(lp[0]), float(np.abs(g).sum())))
""" % ROOT
    out = {}
    for flag in ("1", "0"):
        env = dict(os.environ, POLEE_DEVICE_BUILD=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
        out[flag] = (float(line[2]), float(line[3]))
        assert int(line[1]) == int(flag), line  # (the layout really came from the builder the switch names)
    # the same likelihood through either layout builder (f32 sums in a different order: 1e-5)
    assert abs(out["0"][0] - out["1"][0]) <= 1e-5 * abs(out["1"][0]), out
    assert abs(out["0"][1] - out["1"][1]) <= 1e-4 * abs(out["1"][1]), out
