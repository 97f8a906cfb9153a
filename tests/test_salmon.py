"""Salmon ingest (src/salmon.jl:5-78): parser on CPU; the factored fit from its output on the GPU."""
import gzip
import os

import numpy as np
import pytest

from polee_amd.salmon import load_salmon_likelihood


def _write_salmon_dir(path, names_salmon, classes, efflens, with_weights=True):
    os.makedirs(os.path.join(path, "aux_info"))
    with gzip.open(os.path.join(path, "aux_info", "eq_classes.txt.gz"), "wt") as f:
        f.write("%d\n%d\n" % (len(names_salmon), len(classes)))
        for t in names_salmon:
            f.write(t + "\n")
        for idx, w, k in classes:
            cols = [str(len(idx))] + [str(i) for i in idx] + (["%.9g" % x for x in w] if with_weights else []) + [str(k)]
            f.write("\t".join(cols) + "\n")
    with open(os.path.join(path, "quant.sf"), "w") as f:
        f.write("Name\tLength\tEffectiveLength\tTPM\tNumReads\n")
        for t, e in efflens.items():
            f.write("%s\t%d\t%.3f\t0.0\t0.0\n" % (t, int(e) + 200, e))


def _example(tmp_path, rng, n=40, m=120):
    names_polee = ["tx%03d" % i for i in range(n)]
    names_salmon = list(rng.permutation(names_polee))  # salmon's own order differs from polee's
    classes = []
    for _ in range(m):
        k = int(rng.integers(1, 5))
        idx = sorted(rng.choice(n, k, replace=False).tolist())
        w = rng.dirichlet(np.ones(k)).astype(np.float32)
        classes.append((idx, w, int(rng.integers(1, 50))))
    efflens = {t: float(rng.uniform(100, 3000)) for t in names_polee}
    d = tmp_path / "salmon"
    _write_salmon_dir(str(d), names_salmon, classes, efflens)
    return str(d), names_polee, names_salmon, classes, efflens


def test_salmon_parser_maps_indexes_and_reads_counts(tmp_path):
    rng = np.random.default_rng(50)
    d, names_polee, names_salmon, classes, efflens = _example(tmp_path, rng)
    s = load_salmon_likelihood(d, names_polee)
    assert (s.m, s.n) == (len(classes), len(names_polee))
    np.testing.assert_array_equal(s.ks, [c[2] for c in classes])
    np.testing.assert_allclose(s.efflens, [np.float32("%.3f" % efflens[t]) for t in names_polee], rtol=1e-6)
    import scipy.sparse as sp
    X = sp.csc_matrix((s.nzval, s.rowval.astype(np.int64) - 1, s.colptr.astype(np.int64) - 1), shape=(s.m, s.n)).toarray()
    pos = {t: i for i, t in enumerate(names_polee)}
    for i, (idx, w, _) in enumerate(classes):
        for j, wj in zip(idx, w):
            assert abs(X[i, pos[names_salmon[j]]] - np.float32("%.9g" % wj)) < 1e-7
    assert np.count_nonzero(X) == sum(len(c[0]) for c in classes)
    # error behaviour of the reference
    with pytest.raises(RuntimeError, match="different sets of transcripts"):
        load_salmon_likelihood(d, names_polee[:-1] + ["other"])
    with pytest.raises(RuntimeError, match="Missing likelihood data"):
        load_salmon_likelihood(str(tmp_path / "nowhere"), names_polee)
    d2 = tmp_path / "counts_only"
    _write_salmon_dir(str(d2), names_salmon, classes, efflens, with_weights=False)
    with pytest.raises(RuntimeError, match="Missing likelihood data"):
        load_salmon_likelihood(str(d2), names_polee)


@pytest.mark.gpu
def test_factored_likelihood_from_salmon_output_matches_oracle(tmp_path):
    import polee_amd as P
    from oracle import oracle as O
    rng = np.random.default_rng(51)
    d, names_polee, *_ = _example(tmp_path, rng, n=60, m=400)
    s = load_salmon_likelihood(d, names_polee)
    ctx = P.Context(0)
    sample = s.to_sample(ctx=ctx)
    x = rng.dirichlet(np.ones(s.n)).astype(np.float32)
    lp, g = P.factored_log_likelihood(sample, x)
    so = O.Sample(s.m, s.n, s.colptr, s.rowval, s.nzval)
    lpo, go = so.factored_log_likelihood(s.ks, x)
    assert abs(lp - lpo) <= 1e-4 * abs(lpo)
    np.testing.assert_allclose(g, go, rtol=1e-4, atol=1e-4 * np.abs(go).max())


@pytest.mark.gpu
def test_prep_salmon_file_to_file(tmp_path):
    """`polee prep-salmon` (src/main.jl:723-750) through python -m polee_amd.prep: salmon directory + tree file in,
    prepared sample out; the fit explains the equivalence-class counts better than the starting point."""
    import polee_amd as P
    from polee_amd import h5io, prep
    from conftest import random_tree
    from oracle import oracle as O
    rng = np.random.default_rng(52)
    n = 50
    d, names_polee, *_ = _example(tmp_path, rng, n=n, m=600)
    parents, js = random_tree(n, rng)
    tree_file, ids_file, out_file = str(tmp_path / "tree.h5"), str(tmp_path / "ids.txt"), str(tmp_path / "prep.h5")
    with h5io.File(tree_file, "w") as h:
        h.write("node_parent_idxs", np.ascontiguousarray(parents, np.int32))
        h.write("node_js", np.ascontiguousarray(js, np.int32))
    with open(ids_file, "w") as f:
        f.write("\n".join(names_polee) + "\n")
    assert prep.main(["--salmon", d, "--ptt-tree", tree_file, "--transcript-ids", ids_file, "-o", out_file]) == 0
    got = h5io.read_prepared_sample(out_file)
    assert got["n"] == n and got["m"] == 600
    np.testing.assert_array_equal(got["node_js"], js)
    assert all(np.all(np.isfinite(got[k])) and got[k].shape == (n - 1,) for k in ("mu", "omega", "alpha"))
    # draws from the fit have a higher factored likelihood than draws from the initial approximation
    s = load_salmon_likelihood(d, names_polee)
    so, to = O.Sample(s.m, s.n, s.colptr, s.rowval, s.nzval), O.PTT(parents, js)

    def mean_lp(mu, omega, alpha):
        lps = []
        for k in range(20):
            x = O.sampler_draw(to, mu, np.exp(omega), alpha, O.randn(n - 1, 700 + k))
            lps.append(so.factored_log_likelihood(s.ks, np.maximum(x, 1e-12).astype(np.float32))[0])
        return np.mean(lps)

    zero = np.zeros(n - 1, np.float32)
    assert mean_lp(got["mu"], got["omega"], got["alpha"]) > mean_lp(zero, zero, zero) + 100
    with pytest.raises(SystemExit):
        prep.main(["--salmon", d, "-o", out_file])
