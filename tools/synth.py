"""Synthetic RNA-Seq samples and trees for benchmarks and full-size tests.
Bench/test support; not part of the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libpolee_synth.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            # (several ranks of one job may arrive here together: one builds, the others wait on the lock)
            import fcntl
            with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if not os.path.exists(_LIB):
                    subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_LIB)
        _lib.synth_create.restype = C.c_void_p
        _lib.synth_count.restype = C.c_int64
        _lib.synth_mean_nnz.restype = C.c_double
        _lib.synth_gamma.restype = C.c_double
    return _lib


def make_sample(n, m, mean_nnz=8.0, seed=123456789, dropout=0.0, literal=False):
    """Returns dict(m, n, nnz, tcolptr u64[m+1], trowval u32, tnzval f32, effective_lengths f32[n], gene i32[n]).
    dropout / literal: set diversity (see synth.c) -- per-entry dropout probability; every fragment its own random subset
    of its gene's isoforms (SURVEY 8(d)'s literal wording) instead of one of the gene's <= 12 patterns."""
    L = lib()
    h = C.c_void_p(L.synth_create(C.c_int64(n), C.c_int64(m), C.c_double(mean_nnz), C.c_uint64(seed)))
    if dropout or literal:
        L.synth_set_diversity(h, C.c_double(dropout), C.c_int(1 if literal else 0))
    tcolptr = np.zeros(m + 1, np.uint64)
    nnz = L.synth_count(h, tcolptr.ctypes.data_as(C.c_void_p))
    trowval = np.empty(nnz, np.uint32)
    tnzval = np.empty(nnz, np.float32)
    eff = np.empty(n, np.float32)
    L.synth_fill(h, tcolptr.ctypes.data_as(C.c_void_p), trowval.ctypes.data_as(C.c_void_p),
                 tnzval.ctypes.data_as(C.c_void_p), eff.ctypes.data_as(C.c_void_p))
    gene = np.empty(n, np.int32)
    L.synth_gene_of_transcript(h, gene.ctypes.data_as(C.c_void_p))
    out = dict(m=m, n=n, nnz=int(nnz), tcolptr=tcolptr, trowval=trowval, tnzval=tnzval, effective_lengths=eff,
               gene=gene, num_genes=int(L.synth_num_genes(h)), gamma=float(L.synth_gamma(h)))
    L.synth_free(h)
    return out


def to_csc(s):
    """Xt (CSR) -> the CSC arrays of the likelihood-matrix HDF5 (colptr u64 1-based, rowval u32 1-based, nzval)."""
    m, n, nnz = s["m"], s["n"], s["nnz"]
    colptr = np.zeros(n + 1, np.uint64)
    rowval = np.empty(nnz, np.uint32)
    nzval = np.empty(nnz, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().synth_csr_to_csc(C.c_int64(m), C.c_int64(n), p(s["tcolptr"]), p(s["trowval"]), p(s["tnzval"]), p(colptr),
                           p(rowval), p(nzval))
    return colptr, rowval, nzval


def make_tree(gene, seed=1, kind="hclust"):
    """Serialised tree (node_parent_idxs, node_js; 1-based, DFS pre-order, right child first).
    'hclust': isoforms of a gene form a subtree; gene subtrees are joined by a random recursive
    split that peels single genes off with probability 0.3 (long spines, like hclust output);
    'balanced': perfectly balanced over transcripts in order; 'spine': caterpillar (the reference's
    :sequential tree, hclust.jl:477-489)."""
    rng = np.random.default_rng(seed)
    gene = np.asarray(gene)
    n = gene.size
    parents = np.zeros(2 * n - 1, np.int32)
    js = np.zeros(2 * n - 1, np.int32)
    # work items: (lo, hi, parent, level) over an ordering of leaves; level 0 = genes, 1 = within gene
    if kind == "hclust":
        starts = np.flatnonzero(np.r_[True, gene[1:] != gene[:-1]])
        ends = np.r_[starts[1:], n]
        ng = starts.size
    idx = 0
    stack = [("g", 0, (starts.size if kind == "hclust" else n), 0)]
    while stack:
        typ, lo, hi, par = stack.pop()
        me = idx + 1
        if typ == "g" and kind == "hclust":
            if hi - lo == 1:
                stack.append(("t", int(starts[lo]), int(ends[lo]), par))
                continue
            parents[idx] = par; idx += 1
            if rng.random() < 0.3:
                cut = lo + 1 if rng.random() < 0.5 else hi - 1
            else:
                cut = int(rng.integers(lo + 1, hi))
            stack.append(("g", lo, cut, me)); stack.append(("g", cut, hi, me))
        else:
            if hi - lo == 1:
                parents[idx] = par; js[idx] = lo + 1; idx += 1
                continue
            parents[idx] = par; idx += 1
            if kind == "spine":
                cut = lo + 1
            elif kind == "balanced":
                cut = (lo + hi) // 2
            else:
                cut = int(rng.integers(lo + 1, hi))
            stack.append(("t", lo, cut, me)); stack.append(("t", cut, hi, me))
    assert idx == 2 * n - 1
    return parents, js


def tile_fixture(reps, golden_dir=None, copies=1):
    """REAL-STRUCTURE workload: the reference's likelihood-matrix fixture (tests/golden, m = 19 743 fragments x n = 313
    transcripts, 42 775 non-zeros, 496 distinct transcript sets) tiled block-diagonally `reps` times -- reps = 639 gives
    n ~ 200 k, m ~ 12.6 M with the real distribution of set sizes and run lengths.  Same dict as make_sample (gene =
    one pseudo-gene per 4 transcripts of a block, only used to build a tree).
    copies > 1: every fragment of a block `copies` times -- the same transcript set, the copies' probabilities scaled by
    seeded factors in [0.5, 1.5) -- i.e. the fixture's set structure at a deeper sequencing depth: reps = 639, copies = 9
    gives m = 113.5 M fragments and 246 M non-zeros, BASELINE C2's size."""
    import scipy.sparse as sp
    g = golden_dir or os.path.join(os.path.dirname(_HERE), "tests", "golden")
    d = np.load(os.path.join(g, "mBr_M_6w_1.likelihood-matrix.npz"))
    m0, n0 = int(d["m"].item()), int(d["n"].item())
    X = sp.csc_matrix((d["nzval"].astype(np.float32), d["rowval"].astype(np.int64) - 1, d["colptr"].astype(np.int64) - 1),
                      shape=(m0, n0)).tocsr()
    X.sort_indices()
    if copies > 1:
        rng = np.random.default_rng(20260504)
        X = sp.vstack([X] + [sp.csr_matrix((X.data * rng.uniform(0.5, 1.5, X.nnz).astype(np.float32), X.indices, X.indptr), shape=X.shape)
                             for _ in range(copies - 1)]).tocsr()
        m0 *= copies
    nnz0 = X.nnz
    m, n = m0 * reps, n0 * reps
    ptr0 = X.indptr.astype(np.uint64)
    tcolptr = np.empty(m + 1, np.uint64)
    tcolptr[0] = 1
    trowval = np.empty(nnz0 * reps, np.uint32)
    tnzval = np.empty(nnz0 * reps, np.float32)
    idx0 = X.indices.astype(np.uint32) + 1
    for b in range(reps):
        tcolptr[1 + b * m0:1 + (b + 1) * m0] = ptr0[1:] + np.uint64(b * nnz0 + 1)
        trowval[b * nnz0:(b + 1) * nnz0] = idx0 + np.uint32(b * n0)
        tnzval[b * nnz0:(b + 1) * nnz0] = X.data
    eff = np.tile(d["effective_lengths"].astype(np.float32), reps)
    gene = (np.arange(n) // 4).astype(np.int32)
    return dict(m=m, n=n, nnz=nnz0 * reps, tcolptr=tcolptr, trowval=trowval, tnzval=tnzval, effective_lengths=eff,
                gene=gene, num_genes=int(gene[-1]) + 1, gamma=0.0)
