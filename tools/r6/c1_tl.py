"""C1 (2^20 f32, 1-D, Lorenzo + regression blocks of 128, abs 1e-3) steps for a kernel timeline"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import sz3_amd
from fields import field1d
n = 1 << 20
a = field1d(n, np.float32)
dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(n)
conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
if len(sys.argv) > 1 and sys.argv[1] == "lorenzo":
    conf.regression = 0
if len(sys.argv) > 1 and sys.argv[1] == "default":
    conf.cmprAlgo = sz3_amd.ALGO_INTERP_LORENZO
conf.errorBoundMode = sz3_amd.EB_ABS
conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(n, np.float32)
cap = max(dc.payload_bound(n), dc.payload_bound_conf(conf))
pl = torch.empty(cap, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, st)
torch.cuda.synchronize()
print("C1 %s: %.4f ms/step ratio %.3f" % (sys.argv[1:] or ["composed"], (time.perf_counter() - t0) / 20 * 1e3, a.nbytes / size))
