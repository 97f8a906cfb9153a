import sys,time; sys.path.insert(0,'.')
import numpy as np
import polee_amd as P
from tools import synth
n,m=200000,30000000
smp=synth.make_sample(n,m,8.0,123456789)
colptr,rowval,nzval=synth.to_csc(smp)
t0=time.time(); p,j=P.hclust(m,n,colptr,rowval); print("hclust C2: %.1f s"%(time.time()-t0))
N=len(j); depth=np.zeros(N,np.int64)
for i in range(1,N): depth[i]=depth[p[i]-1]+1
print("max depth",depth.max())
