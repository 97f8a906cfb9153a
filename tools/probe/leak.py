import sys, gc; sys.path.insert(0, '.')
import numpy as np, torch
import polee_amd as P
d = np.load('tests/golden/mBr_M_6w_1.likelihood-matrix.npz'); pr = np.load('tests/golden/mBr_M_6w_1.prep.npz')
m, n = int(d['m'].item()), int(d['n'].item())
torch.cuda.init()
def free():
    gc.collect(); torch.cuda.synchronize(); return torch.cuda.mem_get_info(0)[0]
def run(what):
    ctx = P.Context(0)
    objs = [ctx]
    if 's' in what:
        s = P.RNASeqSample(m, n, d['colptr'], d['rowval'], d['nzval'], d['effective_lengths'], ctx=ctx); objs.append(s)
    if 't' in what:
        t = P.PolyaTreeTransform(pr['node_parent_idxs'], pr['node_js'], ctx=ctx); objs.append(t)
    if 'f' in what:
        fit = P.LikelihoodApproximationFit(s, t, num_steps=3, num_mc_samples=6); fit.run(3); fit.sync(); objs.append(fit)
    if 'c' in what:
        comm = P.Comm(ctx, 1, 0); objs.append(comm)
    if 'g' in what:
        fit2 = P.LikelihoodApproximationFit(s, t, num_steps=2, num_mc_samples=2, comm=comm); fit2.run(2); fit2.sync(); objs.append(fit2)
    if 'r' in what:
        objs.reverse()
    while objs:
        objs.pop(0)
for what in ['stfc', 'stfcg', 'stfcgr', 'stfr']:
    run(what); b = free()
    for _ in range(5): run(what)
    print(repr(what), 'leak per cycle MB', (b - free()) / 5 / 2**20)
