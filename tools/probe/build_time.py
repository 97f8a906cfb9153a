import sys, time, os, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np
from polee_amd import _lib as L
from tools import synth
n,m=200000,30000000
smp=synth.make_sample(n,m,8.0,123456789)
colptr,rowval,nzval=synth.to_csc(smp)
h=C.c_void_p()
t0=time.time()
L.check(L.lib().polee_debug_psell_build(C.c_int64(m),C.c_int64(n),colptr.ctypes.data_as(C.c_void_p),8,L.ptr(rowval,L.u32p),L.ptr(nzval,L.f32p),None,C.byref(h)))
print("total (incl. csc->csr) %.2f s"%(time.time()-t0))
