"""-m gpu: the sparse kernel is compiled with three hidden LLVM backend options (csrc/Makefile); this is the gate that
keeps the build honest on any compiler version: a second library built WITHOUT them (libpolee_hip_untuned.so, `make
untuned`) must give bit-identical results in deterministic mode, at BASELINE's C2 size and on a set-diverse sample that
exercises the masked and mixed streams."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CODE = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import polee_amd as P
    from polee_amd import _lib as L
    from tools import synth
    out = sys.argv[1]
    res = {"version": L.lib().polee_version().decode()}
    ctx = P.Context(0)
    for name, (n, m, kw) in {"c2": (200000, 30000000, {}), "diverse": (20000, 2000000, {"dropout": 0.3})}.items():
        smp = synth.make_sample(n, m, 8.0, seed=123456789, **kw)
        s = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx,
                           xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
        s.set_deterministic(True)
        rng = np.random.default_rng(0)
        x = rng.gamma(0.3, size=(6, n)).astype(np.float32) + np.float32(1e-7)
        x /= x.sum(axis=1, keepdims=True)
        lp, g = s.log_likelihood(np.clip(x, np.float32(1e-10), 1))
        res[name + "_lp"], res[name + "_g"] = lp, g
        del s
    np.savez(out, **res)
""") % ROOT


def _run(lib, out):
    env = dict(os.environ)
    if lib:
        env["POLEE_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CODE, out], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def test_tuned_and_untuned_builds_agree_bit_for_bit(tmp_path):
    untuned = os.path.join(ROOT, "polee_amd", "csrc", "libpolee_hip_untuned.so")
    if not os.path.exists(untuned):
        pytest.skip("libpolee_hip_untuned.so is not built (make -C polee_amd/csrc untuned)")
    a = _run(None, str(tmp_path / "tuned.npz"))
    b = _run(untuned, str(tmp_path / "untuned.npz"))
    va, vb = str(a["version"]), str(b["version"])
    print("product build:", va)
    print("gate build:   ", vb)
    assert "gate build" in vb and "gate build" not in va
    for key in ("c2_lp", "c2_g", "diverse_lp", "diverse_g"):
        assert np.array_equal(a[key], b[key]), key
    assert np.isfinite(a["c2_g"]).all() and (a["c2_lp"] < 0).all()
