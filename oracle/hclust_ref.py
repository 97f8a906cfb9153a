"""CPU restatement of the reference's tree construction (test infrastructure only, like the rest of oracle/):
hclust (src/hclust.jl:193-319) and order_nodes (src/hclust.jl:361-389) in plain Python, for small cases.

Julia details followed: sortperm is stable (:212); HClustEdge keeps the similarity as Float32 (:80-84) and the
heaps compare with strict < / > (:87-110); DataStructures.jl's BinaryHeap = append + percolate up, pop = last to
the root + percolate down; MultiDict keeps neighbours in insertion order.  Not reproducible from Julia: the Dict
iteration order of the remaining components (:245) -- ascending node id is used here."""
import numpy as np

K_NEIGHBOURS = 25  # hclust.jl:200


class _Heap:
    def __init__(self, before):
        self.xs, self.before = [], before

    def push(self, x):
        xs = self.xs
        xs.append(x)
        i = len(xs)
        while i > 1:
            j = i // 2
            if self.before(x, xs[j - 1]):
                xs[i - 1] = xs[j - 1]
                i = j
            else:
                break
        xs[i - 1] = x

    def pop(self):
        xs = self.xs
        top, y = xs[0], xs.pop()
        n = len(xs)
        if n:
            i = 1
            while True:
                l, r = 2 * i, 2 * i + 1
                if l > n:
                    break
                j = l if (r > n or self.before(xs[l - 1], xs[r - 1])) else r
                if self.before(xs[j - 1], y):
                    xs[i - 1] = xs[j - 1]
                    i = j
                else:
                    break
            xs[i - 1] = y
        return top


def _rel_intersection(a, b):  # hclust.jl:143-152
    if len(a) == 0 and len(b) == 0:
        return 0.0
    inter = len(np.intersect1d(a, b, assume_unique=True))
    return inter / (len(a) + len(b) - inter)


def hclust(m, n, colptr, rowval):
    """X (m x n) in CSC, 1-based -> (node_parent_idxs, node_js), int32 [2n-1] each."""
    colptr = np.asarray(colptr).astype(np.int64)
    rowval = np.asarray(rowval).astype(np.int64)
    med = np.zeros(n, np.int64)
    for j in range(n):
        if colptr[j] != colptr[j + 1]:
            med[j] = rowval[(colptr[j] + colptr[j + 1]) // 2 - 1]
    idxs = np.argsort(med, kind="stable")
    nodes = {}      # id -> (j, left, right)
    read_sets = {}
    for j in range(1, n + 1):
        t = idxs[j - 1]
        nodes[j] = (int(t) + 1, None, None)
        read_sets[j] = rowval[colptr[t] - 1:colptr[t + 1] - 1]
    queue = _Heap(lambda a, b: a[2] > b[2])
    neighbours = {j: [] for j in range(1, n + 1)}
    for j1 in range(1, n + 1):
        for j2 in range(j1 + 1, min(j1 + K_NEIGHBOURS, n) + 1):
            sim = _rel_intersection(read_sets[j1], read_sets[j2])
            if sim > 0:
                queue.push((j1, j2, np.float32(sim)))
            neighbours[j1].append(j2)
            neighbours[j2].append(j1)
    nxt = n + 1
    deleted = set()
    tree = dict(nodes)
    while queue.xs:
        j1, j2, sim = queue.pop()
        if j1 in deleted or j2 in deleted:
            continue
        k = nxt
        nxt += 1
        read_sets[k] = np.union1d(read_sets[j1], read_sets[j2])
        tree[k] = (0, j1, j2)
        nodes[k] = tree[k]
        for j in (j1, j2):
            del nodes[j], read_sets[j]
            deleted.add(j)
        neighbours[k] = []
        for ja, jb in ((j1, j2), (j2, j1)):
            for l in list(neighbours[ja]):
                if l == jb or l in deleted:
                    continue
                s2 = _rel_intersection(read_sets[l], read_sets[k])
                if s2 != 0:
                    queue.push((l, k, np.float32(s2)))
                neighbours[l].append(k)
                neighbours[k].append(l)
    rest = _Heap(lambda a, b: a[1] < b[1])
    for j in sorted(nodes):
        rest.push((j, 1 + len(read_sets[j])))
    while len(rest.xs) > 1:
        a, b = rest.pop(), rest.pop()
        k = nxt
        nxt += 1
        tree[k] = (0, a[0], b[0])
        rest.push((k, a[1] + b[1]))
    root = rest.pop()[0]
    parents, js, parent_of, stack = [], [], {root: 0}, [root]
    while stack:
        v = stack.pop()
        parents.append(parent_of[v])
        js.append(tree[v][0])
        if tree[v][0] == 0:
            parent_of[tree[v][1]] = parent_of[tree[v][2]] = len(parents)
            stack.append(tree[v][1])
            stack.append(tree[v][2])
    return np.array(parents, np.int32), np.array(js, np.int32)


# ---- the parallel variant (polee_hclust_parallel, polee_amd/csrc/hclust.cpp "rounds of mutually-best merges") --------
# Not the reference's algorithm: the same joining rule with a local merge order.  Restated here, sequentially, from
# its definition so that the threaded C++ has something to be node-for-node equal to.
_M64 = (1 << 64) - 1


def _edge_pri(a, b, sim32):
    """Priority of an edge: (similarity bits << 32 | hash(lo, hi)), then smaller lo, then smaller hi, first."""
    lo, hi = (a, b) if a < b else (b, a)
    h = ((lo * 0x9E3779B97F4A7C15) & _M64) ^ ((hi * 0xC2B2AE3D27D4EB4F + 0x165667B19E3779F9) & _M64)
    h ^= h >> 29
    h = (h * 0xBF58476D1CE4E5B9) & _M64
    h ^= h >> 32
    key = (int(np.float32(sim32).view(np.uint32)) << 32) | (h & 0xFFFFFFFF)
    return (key, -lo, -hi)


def hclust_rounds(m, n, colptr, rowval):
    colptr = np.asarray(colptr).astype(np.int64)
    rowval = np.asarray(rowval).astype(np.int64)
    med = np.zeros(n, np.int64)
    for j in range(n):
        if colptr[j] != colptr[j + 1]:
            med[j] = rowval[(colptr[j] + colptr[j + 1]) // 2 - 1]
    idxs = np.argsort(med, kind="stable")
    tree, read_sets, adj = {}, {}, {}
    for j in range(1, n + 1):
        t = idxs[j - 1]
        tree[j] = (int(t) + 1, None, None)
        read_sets[j] = rowval[colptr[t] - 1:colptr[t + 1] - 1]
        adj[j] = {}
    for j1 in range(1, n + 1):
        for j2 in range(j1 + 1, min(j1 + K_NEIGHBOURS, n) + 1):
            sim = np.float32(_rel_intersection(read_sets[j1], read_sets[j2]))
            if sim > 0:  # neighbours without a common read are not listed
                adj[j1][j2] = sim
                adj[j2][j1] = sim
    size = {j: len(read_sets[j]) for j in read_sets}
    nxt = n + 1
    while True:
        best = {}
        for a, nb in adj.items():
            if nb:
                best[a] = max(nb, key=lambda l: _edge_pri(a, l, nb[l]))
        pairs = sorted({(min(a, b), max(a, b)) for a, b in best.items() if best.get(b) == a},
                       key=lambda e: _edge_pri(e[0], e[1], adj[e[0]][e[1]]), reverse=True)
        if not pairs:
            break
        into = {}
        for lo, hi in pairs:
            into[lo] = into[hi] = nxt
            tree[nxt] = (0, lo, hi)
            nxt += 1
        new_sets, new_cands = {}, {}
        for lo, hi in pairs:
            k = into[lo]
            new_sets[k] = np.union1d(read_sets[lo], read_sets[hi])
            new_cands[k] = sorted({into.get(l, l) for half in (lo, hi) for l in adj[half] if l not in (lo, hi)})
        for lo, hi in pairs:
            for j in (lo, hi):
                for l in adj[j]:
                    if l in adj and l not in into:
                        adj[l].pop(j, None)
                del adj[j], read_sets[j], size[j]
        read_sets.update(new_sets)
        for k, cs in new_cands.items():
            adj[k] = {}
            size[k] = len(read_sets[k])
        for k, cs in new_cands.items():
            for l in cs:
                sim = np.float32(_rel_intersection(read_sets[l], read_sets[k]))
                if sim > 0:
                    adj[k][l] = sim
                    adj[l][k] = sim
    rest = _Heap(lambda a, b: a[1] < b[1])
    for j in sorted(adj):
        rest.push((j, 1 + size[j]))
    while len(rest.xs) > 1:
        a, b = rest.pop(), rest.pop()
        tree[nxt] = (0, a[0], b[0])
        rest.push((nxt, a[1] + b[1]))
        nxt += 1
    root = rest.pop()[0]
    parents, js, parent_of, stack = [], [], {root: 0}, [root]
    while stack:
        v = stack.pop()
        parents.append(parent_of[v])
        js.append(tree[v][0])
        if tree[v][0] == 0:
            parent_of[tree[v][1]] = parent_of[tree[v][2]] = len(parents)
            stack.append(tree[v][1])
            stack.append(tree[v][2])
    return np.array(parents, np.int32), np.array(js, np.int32)
