import sys; sys.path.insert(0, '.')
import numpy as np
import polee_amd as P
d = np.load('tests/golden/mBr_M_6w_1.likelihood-matrix.npz'); pr = np.load('tests/golden/mBr_M_6w_1.prep.npz')
m, n = int(d['m'].item()), int(d['n'].item())
ctx = P.Context(0)
s = P.RNASeqSample(m, n, d['colptr'], d['rowval'], d['nzval'], d['effective_lengths'], ctx=ctx)
t = P.PolyaTreeTransform(pr['node_parent_idxs'], pr['node_js'], ctx=ctx)
for seed in (1, 2):
    got = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, seed=seed)
    for k in ('mu', 'omega', 'alpha'):
        a, b = got[k], pr[k]
        print(seed, k, 'corr %.4f' % np.corrcoef(a, b)[0, 1], 'median abs diff %.4f' % np.median(np.abs(a - b)), 'ref std %.3f' % b.std())
g1 = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, seed=1)
g2 = P.approximate_likelihood(P.LogitSkewNormalPTTApprox(), s, t, seed=2)
for k in ('mu', 'omega', 'alpha'):
    print('ours seed1 vs seed2', k, 'corr %.4f' % np.corrcoef(g1[k], g2[k])[0, 1], 'median abs diff %.4f' % np.median(np.abs(g1[k] - g2[k])))
