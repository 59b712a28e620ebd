#!/usr/bin/env python3
"""the C1 step (1-D 2^20 f32, ALGO_LORENZO_REG defaults, abs 1e-3) in a loop, for a kernel trace (tools/c1_timeline.sh); DEC=1: the decoder"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field1d
dev = torch.device("cuda:0")
n = int(os.environ.get("N", 1 << 20))
a = field1d(n, np.float32)
d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(n)
if not os.environ.get("DEFAULT_ALGO"): conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(n, np.float32)
cap = max(dc.payload_bound(n), dc.payload_bound_conf(conf))
d_pl = torch.empty(cap, dtype=torch.uint8, device=dev)
d_out = torch.empty(n, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def comp():
    dc.stage1(conf, d_in.data_ptr(), st); dc.stage2(d_pl.data_ptr(), cap, st); return dc.finish(st)
for _ in range(4): size = comp()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): size = comp()
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / 20
for _ in range(3): dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 20
print("C1 step %.4f ms, decompress %.4f ms, ratio %.3f" % (tc * 1e3, td * 1e3, a.nbytes / size))
