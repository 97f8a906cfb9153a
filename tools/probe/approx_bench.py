import os, sys, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
torch.cuda.init()  # (torch's HIP runtime must come up before libpolee_hip's in a shared process)
import polee_amd as P
from tools import synth
n = 200000
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(0)
smp = synth.make_sample(n, 1000000, 8.0, 1)
par, js = synth.make_tree(smp['gene'], 1)
l, r, f = P.make_inverse_ptt_params(par, js)
L_, R_, F_ = (np.tile(a, (S, 1)) for a in (l, r, f))
eff = np.tile(smp['effective_lengths'], (S, 1)).astype(np.float32)
mu = rng.normal(0, 2, (S, n - 1)).astype(np.float32)
sigma = np.exp(rng.normal(-1, 1, (S, n - 1))).astype(np.float32)
alpha = rng.normal(0, .3, (S, n - 1)).astype(np.float32)
ctx = P.Context(0)
ap = P.RNASeqApproxLikelihood(dict(efflen=eff, la_mu=mu, la_sigma=sigma, la_alpha=alpha, left_index=L_, right_index=R_, leaf_index=F_), ctx=ctx)
x = rng.normal(0, 2, (S, n)).astype(np.float32)
for want in (False, True):
    ap.log_prob(x, want_grad=want)
    t0 = time.perf_counter()
    for _ in range(20): ap.log_prob(x, want_grad=want)
    dt = (time.perf_counter() - t0) / 20
    print("S=%d n=%d host-API log_prob grad=%s: %.2f ms per call" % (S, n, want, dt * 1e3))
# device-only timing through the C API (torch tensors as device buffers)
import ctypes as C
from polee_amd import _lib as L
xd = torch.tensor(x, device='cuda'); lpd = torch.zeros(S, device='cuda'); gd = torch.zeros(S, n, device='cuda')
torch.cuda.synchronize()
f = L.lib().polee_approx_logprob_device
f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for want in (False, True):
    for _ in range(3): L.check(f(ap._h, xd.data_ptr(), lpd.data_ptr(), gd.data_ptr() if want else None))
    ctx.synchronize(); ctx.timer_start()
    for _ in range(50): L.check(f(ap._h, xd.data_ptr(), lpd.data_ptr(), gd.data_ptr() if want else None))
    ms = ctx.timer_stop() / 50
    print("S=%d device log_prob grad=%s: %.3f ms per call = %.1f calls/s (%.1f MB/sample moved if ~16 MB)" % (S, want, ms, 1e3 / ms, 16.0))
