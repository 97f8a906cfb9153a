import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _oracle_threads():
    """The oracle forks an OpenMP team per call; on a 256-thread host the fixture-sized tests (313 transcripts) would
    spend their time in fork/join.  Tests at BASELINE sizes raise the count themselves."""
    try:
        from oracle import oracle as O
        O.set_num_threads(min(8, O.physical_cores()))
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def lm_fixture():
    """Reference-produced likelihood matrix (test/dataset/mBr_M_6w_1.likelihood-matrix.h5)."""
    a = np.load(os.path.join(GOLDEN, "mBr_M_6w_1.likelihood-matrix.npz"))
    return dict(m=int(a["m"][0]), n=int(a["n"][0]), colptr=a["colptr"], rowval=a["rowval"],
                nzval=a["nzval"], effective_lengths=a["effective_lengths"])


@pytest.fixture(scope="session")
def prep_fixture():
    """Reference-produced fitted approximation (test/dataset/mBr_M_6w_1.prep.h5)."""
    a = np.load(os.path.join(GOLDEN, "mBr_M_6w_1.prep.npz"))
    return {k: a[k] for k in a.files}


def random_tree(n, rng, kind="random"):
    """Serialised tree (node_parent_idxs, node_js), DFS pre-order, right child first,
    as written by the reference (src/hclust.jl:361-389 order_nodes).
    kind: 'random' (random merges), 'spine' (caterpillar, like list_nodes hclust.jl:477-489),
    'balanced'."""
    # build nested tuples: leaf = int (1-based transcript id), internal = (left, right)
    perm = list(rng.permutation(n) + 1)
    if kind == "spine":
        t = perm[0]
        for j in perm[1:]:
            t = (t, j) if rng.random() < 0.5 else (j, t)
    elif kind == "balanced":
        level = perm
        while len(level) > 1:
            nxt = [(level[i], level[i + 1]) for i in range(0, len(level) - 1, 2)]
            if len(level) % 2:
                nxt.append(level[-1])
            level = nxt
        t = level[0]
    else:
        pool = perm
        while len(pool) > 1:
            i = int(rng.integers(len(pool)))
            a = pool.pop(i)
            j = int(rng.integers(len(pool)))
            b = pool.pop(j)
            pool.append((a, b))
        t = pool[0]
    parents, js = [], []
    stack = [(t, 0)]
    while stack:
        node, par = stack.pop()
        idx = len(parents) + 1
        parents.append(par)
        if isinstance(node, tuple):
            js.append(0)
            stack.append((node[0], idx))  # left pushed first ...
            stack.append((node[1], idx))  # ... right popped first
        else:
            js.append(int(node))
    return np.array(parents, np.int32), np.array(js, np.int32)
