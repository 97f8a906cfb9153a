import sys, gc; sys.path.insert(0, '.')
import numpy as np, torch
import polee_amd as P
d = np.load('tests/golden/mBr_M_6w_1.likelihood-matrix.npz'); pr = np.load('tests/golden/mBr_M_6w_1.prep.npz')
m, n = int(d['m'].item()), int(d['n'].item())
torch.cuda.init()
def free():
    gc.collect(); torch.cuda.synchronize(); return torch.cuda.mem_get_info(0)[0]
def cycle(verbose):
    f0 = free()
    ctx = P.Context(0)
    s = P.RNASeqSample(m, n, d['colptr'], d['rowval'], d['nzval'], d['effective_lengths'], ctx=ctx)
    t = P.PolyaTreeTransform(pr['node_parent_idxs'], pr['node_js'], ctx=ctx)
    fit = P.LikelihoodApproximationFit(s, t, num_steps=3, num_mc_samples=6); fit.run(3); fit.sync()
    f1 = free()
    comm = P.Comm(ctx, 1, 0)
    f2 = free()
    del comm
    f3 = free()
    del fit
    f4 = free()
    del s, t, ctx
    f5 = free()
    if verbose:
        print("MB: objects %.1f  comm +%.1f  after comm del %+.1f  after fit del %+.1f  end vs start %+.1f" % (
            (f0 - f1) / 2**20, (f1 - f2) / 2**20, (f3 - f1) / 2**20, (f4 - f1) / 2**20, (f5 - f0) / 2**20))
cycle(False)
for _ in range(3): cycle(True)
