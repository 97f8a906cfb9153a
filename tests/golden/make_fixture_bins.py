#!/usr/bin/env python3
"""Convert the reference's two HDF5 test fixtures into flat little-endian binaries.

The fixtures are DATA produced by one run of the reference
(/root/reference/test/dataset/Snakefile:96-110):
  mBr_M_6w_1.likelihood-matrix.h5  (written by src/rnaseq_sample.jl:505-519)
  mBr_M_6w_1.prep.h5               (written by src/likelihood-approximation.jl:61-87)
They are the only reference-produced numbers in this repo.  h5py is not
installed in the build container, so the datasets are dumped with the HDF5
command line tool (h5dump -b LE) and concatenated into one .npz per file.

Run in the build container only (needs /root/reference and /opt/conda/bin/h5dump):
    python tests/golden/make_fixture_bins.py
"""
import json, os, subprocess, sys, tempfile
import numpy as np

H5DUMP = "/opt/conda/bin/h5dump"
SRC = "/root/reference/test/dataset"
HERE = os.path.dirname(os.path.abspath(__file__))


def dump(h5, name, dtype):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.check_call([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", tmp.name, h5],
                              stdout=subprocess.DEVNULL)
        return np.fromfile(tmp.name, dtype=dtype)


def attrs(h5, group):
    out = subprocess.check_output([H5DUMP, "-A", "-g", group, h5], text=True)
    res, cur = {}, None
    for line in out.splitlines():
        s = line.strip()
        if s.startswith("ATTRIBUTE"):
            cur = s.split('"')[1]
        elif s.startswith("(0):") and cur is not None:
            v = s[4:].strip()
            res[cur] = v.strip('"') if v.startswith('"') else int(v)
            cur = None
    return res


def main():
    lm = os.path.join(SRC, "mBr_M_6w_1.likelihood-matrix.h5")
    pr = os.path.join(SRC, "mBr_M_6w_1.prep.h5")
    np.savez_compressed(
        os.path.join(HERE, "mBr_M_6w_1.likelihood-matrix.npz"),
        m=dump(lm, "m", "<i8"), n=dump(lm, "n", "<i8"),
        colptr=dump(lm, "colptr", "<u4"), rowval=dump(lm, "rowval", "<u4"),
        nzval=dump(lm, "nzval", "<f4"),
        effective_lengths=dump(lm, "effective_lengths", "<f4"))
    np.savez_compressed(
        os.path.join(HERE, "mBr_M_6w_1.prep.npz"),
        m=dump(pr, "m", "<i8"), n=dump(pr, "n", "<i8"),
        mu=dump(pr, "mu", "<f4"), omega=dump(pr, "omega", "<f4"),
        alpha=dump(pr, "alpha", "<f4"),
        node_parent_idxs=dump(pr, "node_parent_idxs", "<i4"),
        node_js=dump(pr, "node_js", "<i4"),
        effective_lengths=dump(pr, "effective_lengths", "<f4"))
    with open(os.path.join(HERE, "mBr_M_6w_1.prep.metadata.json"), "w") as f:
        json.dump(attrs(pr, "/metadata"), f, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
