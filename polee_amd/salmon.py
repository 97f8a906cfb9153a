"""Salmon ingest (src/salmon.jl:5-78): the factored likelihood matrix of `salmon quant -d` output.

`load_salmon_likelihood(salmon_dir, transcript_ids)` reads aux_info/eq_classes.txt.gz (equivalence classes with
their per-transcript conditional probabilities and read counts) and quant.sf (effective lengths) and returns the
arguments of `RNASeqSample(..., ks=...)`, the device-resident input of the factored likelihood
(src/likelihood.jl:59-85) and of the factored fit (src/likelihood-approximation.jl:248-392).  Host-side parsing only.
"""
import gzip
import os

import numpy as np


class SalmonLikelihood:
    """X (m equivalence classes x n transcripts) in the reference's CSC arrays (1-based), ks, efflens."""

    def __init__(self, m, n, colptr, rowval, nzval, ks, efflens):
        self.m, self.n, self.colptr, self.rowval, self.nzval, self.ks, self.efflens = m, n, colptr, rowval, nzval, ks, efflens

    def to_sample(self, ctx=None):
        from .core import RNASeqSample
        return RNASeqSample(self.m, self.n, self.colptr, self.rowval, self.nzval, self.efflens, ks=self.ks, ctx=ctx)


def load_salmon_likelihood(salmon_dir, transcript_ids):
    """load_salmon_likelihood (src/salmon.jl:5-78).  `transcript_ids`: the transcripts in polee's order (the order of
    the tree's leaves ids); salmon's own indexes are mapped onto it."""
    import scipy.sparse as sp
    tid_map = {tid: i for i, tid in enumerate(transcript_ids)}  # 0-based here
    eqc_filename = os.path.join(salmon_dir, "aux_info", "eq_classes.txt.gz")
    if not os.path.isfile(eqc_filename):
        raise RuntimeError("Missing likelihood data. Please run salmon quand with '-d'")
    with gzip.open(eqc_filename, "rt") as stream:
        n = int(stream.readline())
        m = int(stream.readline())
        salmon_transcript_ids = [stream.readline().rstrip("\n") for _ in range(n)]
        if set(salmon_transcript_ids) != set(transcript_ids):
            raise RuntimeError("'salmon index' and 'polee fit-tree' were used with different sets of transcripts.\n"
                               "You may need to run 'salmon index' with '--keepDuplicates'.")
        efflens = np.zeros(n, np.float32)
        with open(os.path.join(salmon_dir, "quant.sf")) as quant:
            quant.readline()  # header
            for line in quant:
                row = line.rstrip("\n").split("\t")
                efflens[tid_map[row[0]]] = np.float32(row[2])
        to_polee = np.array([tid_map[t] for t in salmon_transcript_ids], np.int64)
        I, J, V = [], [], []
        ks = np.empty(m, np.int64)
        for i in range(m):
            row = stream.readline().rstrip("\n").split("\t")
            nval = int(row[0])
            if len(row) < 2 + 2 * nval:  # counts only: salmon ran without -d
                raise RuntimeError("Missing likelihood data. Please run salmon quand with '-d'")
            ks[i] = int(row[1 + 2 * nval])
            I.extend([i] * nval)
            J.extend(to_polee[[int(t) for t in row[1:1 + nval]]])
            V.extend(np.float32(w) for w in row[1 + nval:1 + 2 * nval])
    X = sp.coo_matrix((np.asarray(V, np.float32), (np.asarray(I, np.int64), np.asarray(J, np.int64))),
                      shape=(m, n)).tocsc()  # duplicates are summed, as Julia's sparse(I, J, V, m, n) does
    X.sort_indices()
    return SalmonLikelihood(m, n, (X.indptr + 1).astype(np.uint32), (X.indices + 1).astype(np.uint32),
                            X.data.astype(np.float32), ks, efflens)
