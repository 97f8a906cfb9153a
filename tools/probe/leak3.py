import sys, gc; sys.path.insert(0, '.')
import numpy as np, torch
import polee_amd as P
d = np.load('tests/golden/mBr_M_6w_1.likelihood-matrix.npz'); pr = np.load('tests/golden/mBr_M_6w_1.prep.npz')
m, n = int(d['m'].item()), int(d['n'].item())
torch.cuda.init()
def free():
    gc.collect(); torch.cuda.synchronize(); return torch.cuda.mem_get_info(0)[0]
def cycle():
    ctx = P.Context(0)
    s = P.RNASeqSample(m, n, d['colptr'], d['rowval'], d['nzval'], d['effective_lengths'], ctx=ctx)
    t = P.PolyaTreeTransform(pr['node_parent_idxs'], pr['node_js'], ctx=ctx)
    fit = P.LikelihoodApproximationFit(s, t, num_steps=3, num_mc_samples=6); fit.run(3); fit.sync()
    comm = P.Comm(ctx, 1, 0)
    fit2 = P.LikelihoodApproximationFit(s, t, num_steps=2, num_mc_samples=2, comm=comm); fit2.run(2); fit2.sync()
    del ctx, s, t
    del comm
    del fit2, fit
cycle(); b = free()
for i in range(1, 31):
    cycle()
    if i in (1, 2, 5, 10, 20, 30): print(i, "cycles: deficit MB %.1f" % ((b - free()) / 2**20))
