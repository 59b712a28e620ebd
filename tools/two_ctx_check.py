import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, sz3_amd
from fields import field3d
S=256; dev=torch.device("cuda:0")
a=torch.from_numpy(field3d((S,S,S))).to(dev)
conf = sz3_amd.Config(S,S,S); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = 1e-3
n=S**3
d0=sz3_amd.DeviceCompressor(n,np.float32); d1=sz3_amd.DeviceCompressor(n,np.float32)
cap=d0.payload_bound(n); p0=torch.empty(cap,dtype=torch.uint8,device=dev); p1=torch.empty(cap,dtype=torch.uint8,device=dev)
for k in range(4):
    d0.compress(conf,a.data_ptr(),p0.data_ptr(),cap,0); d1.compress(conf,a.data_ptr(),p1.data_ptr(),cap,0)
    print(k, d0.spec_stats(), d1.spec_stats())
s=torch.cuda.Stream()
for k in range(3):
    d1.compress(conf,a.data_ptr(),p1.data_ptr(),cap,s.cuda_stream); print("stream", d1.spec_stats())
