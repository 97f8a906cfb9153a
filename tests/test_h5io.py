"""On-disk contracts (prep / likelihood-matrix HDF5) through polee_amd.h5io: round trips, the reference's own
fixture files when the reference tree is mounted (build container only), and the model-entry loader."""
import os

import numpy as np
import pytest

from polee_amd import h5io

REF = "/root/reference/test/dataset"

try:
    h5io.lib()
    HAVE_H5 = True
except Exception:  # libhdf5 missing: nothing to test on this box
    HAVE_H5 = False
pytestmark = pytest.mark.skipif(not HAVE_H5, reason="libhdf5 not available")


def test_likelihood_matrix_roundtrip(tmp_path, lm_fixture):
    f = lm_fixture
    fn = str(tmp_path / "x.likelihood-matrix.h5")
    h5io.write_likelihood_matrix(fn, f["m"], f["n"], f["colptr"], f["rowval"], f["nzval"], f["effective_lengths"],
                                 metadata={"gfffilename": "annotations.gff3"})
    r = h5io.read_likelihood_matrix(fn)
    assert r["m"] == f["m"] and r["n"] == f["n"]
    for k in ("colptr", "rowval", "nzval", "effective_lengths"):
        np.testing.assert_array_equal(r[k], f[k])


def test_prepared_sample_roundtrip_and_version_check(tmp_path, prep_fixture):
    p = prep_fixture
    fn = str(tmp_path / "s.prep.h5")
    h5io.write_approximation(fn, 19743, 313, p["effective_lengths"], p, gfffilename="annotations.gff3",
                             gffhash=b"\x01\x02\x03", fafilename="genome.fa", fahash=b"\x04")
    r = h5io.read_prepared_sample(fn)
    assert r["n"] == 313 and r["m"] == 19743 and r["metadata"]["version"] == 2
    assert r["metadata"]["approximation"] == "Polee.LogitSkewNormalPTTApprox"
    for k in ("mu", "omega", "alpha", "node_parent_idxs", "node_js", "effective_lengths"):
        np.testing.assert_array_equal(r[k], p[k])
    # a file without the version attribute is rejected like likelihood-approximation.jl:94-101
    bad = str(tmp_path / "bad.h5")
    with h5io.File(bad, "w") as f:
        f.write("n", np.int64(3))
        f.create_group("metadata")
    with h5io.File(bad) as f, pytest.raises(RuntimeError, match="older"):
        h5io.check_prepared_sample_version(f, bad)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_reads_the_reference_fixture_files(lm_fixture, prep_fixture):
    """The files the reference itself wrote are read identically to the h5dump-converted npz fixtures."""
    r = h5io.read_likelihood_matrix(os.path.join(REF, "mBr_M_6w_1.likelihood-matrix.h5"))
    assert r["m"] == lm_fixture["m"] and r["n"] == lm_fixture["n"]
    for k in ("colptr", "rowval", "nzval", "effective_lengths"):
        np.testing.assert_array_equal(r[k], lm_fixture[k])
    p = h5io.read_prepared_sample(os.path.join(REF, "mBr_M_6w_1.prep.h5"))
    for k in ("mu", "omega", "alpha", "node_parent_idxs", "node_js", "effective_lengths"):
        np.testing.assert_array_equal(p[k], prep_fixture[k])
    assert p["metadata"]["version"] == 2 and p["metadata"]["gfffilename"] == "annotations.gff3"


def test_read_specification_and_loader_without_device(tmp_path, prep_fixture):
    from polee_amd.estimate import read_specification, load_samples_hdf5
    p = prep_fixture
    files = []
    for i in range(3):
        fn = str(tmp_path / ("s%d.prep.h5" % i))
        q = dict(p)
        q["mu"] = p["mu"] + np.float32(i)
        h5io.write_approximation(fn, 19743, 313, p["effective_lengths"], q, gffhash=b"abc")
        files.append(fn)
    spec = {"samples": [{"name": "s%d" % i, "file": files[i], "factors": {"tissue": "brain", "rep": i}} for i in range(3)]}
    fns, names, factors = read_specification(spec)
    assert fns == files and names == ["s0", "s1", "s2"] and factors[2] == {"tissue": "brain", "rep": "2"}
    assert read_specification({"samples": [{"name": "a"}]})[0] == ["a.likelihood.h5"]  # estimate.jl:12 default suffix
    ls = load_samples_hdf5(files, 313, gffhash=b"abc", using_device=False)
    assert ls.la_mu_values.shape == (3, 312) and ls.left_index.shape == (3, 625)
    np.testing.assert_allclose(ls.la_sigma_values[1], np.exp(p["omega"]), rtol=1e-6)
    np.testing.assert_array_equal(ls.la_mu_values[2], p["mu"] + np.float32(2))
    with pytest.raises(ValueError, match="not the same"):
        load_samples_hdf5(files, 313, gffhash=b"zzz", using_device=False)
    with pytest.raises(ValueError, match="different number"):
        load_samples_hdf5(files, 314, using_device=False)


@pytest.mark.gpu
def test_loader_builds_device_likelihood(tmp_path, prep_fixture):
    """load_samples_from_specification on the device: the 7 variables of create_tensorflow_variables!, and x0 = the mean of
    30 draws as estimate.jl:436-455 draws them, compared with the oracle's x0_draw on the same noise."""
    from oracle import oracle as O
    from polee_amd.estimate import load_samples_hdf5, load_samples_from_specification
    p = prep_fixture
    files = []
    for i in range(2):
        fn = str(tmp_path / ("s%d.prep.h5" % i))
        q = dict(p)
        q["mu"] = p["mu"] + np.float32(0.25 * i)
        h5io.write_approximation(fn, 19743, 313, p["effective_lengths"], q)
        files.append(fn)
    ls = load_samples_from_specification({"samples": [{"name": "a", "file": files[0]}, {"name": "b", "file": files[1]}]}, 313)
    assert set(ls.variables) >= {"efflen", "la_mu", "la_sigma", "la_alpha", "left_index", "right_index", "leaf_index"}
    assert ls.x0_values.shape == (2, 313) and np.allclose(ls.x0_values.sum(axis=1), 1, atol=1e-3)
    lp = ls.variables["approx"].log_prob(np.log(ls.x0_values))
    assert np.isfinite(lp).all()
    # the same loader with supplied noise against the oracle, draw by draw
    N = 30
    noise = np.stack([np.stack([O.randn(312, 100 * i + d) for d in range(N)]) for i in range(2)])
    ls2 = load_samples_hdf5(files, 313, num_init_draws=N, init_noise=noise)
    to = O.PTT(p["node_parent_idxs"], p["node_js"])
    for i in range(2):
        mu = p["mu"] + np.float32(0.25 * i)
        ref = np.zeros(313)
        for d in range(N):
            ref += O.x0_draw(to, mu, np.exp(p["omega"]), p["alpha"], p["effective_lengths"], noise[i, d])
        np.testing.assert_allclose(ls2.x0_values[i], ref / N, rtol=2e-5, atol=1e-12)
    # device-RNG initial values are the same statistic: close to the supplied-noise mean on the expressed transcripts
    big = ls2.x0_values[0] > 1e-3
    assert np.abs(np.log(ls.x0_values[0][big] / ls2.x0_values[0][big])).max() < 1.0


def test_prep_tree_method_auto_names_the_tree_it_builds():
    """python -m polee_amd.prep --tree-method auto (the default): the reference's merge order for small annotations, the rounds
    variant above a stated size, and a note that says which -- an explicit choice is passed through untouched."""
    from polee_amd import prep
    assert prep.resolve_tree_method("cluster", 10 ** 6) == ("cluster", None)
    tm, note = prep.resolve_tree_method("auto", prep.AUTO_EXACT_MAX_N)
    assert tm == "cluster" and "reference" in note
    tm, note = prep.resolve_tree_method("auto", prep.AUTO_EXACT_MAX_N + 1)
    assert tm == "cluster_auto" and "ROUNDS" in note and "--tree-method cluster" in note
