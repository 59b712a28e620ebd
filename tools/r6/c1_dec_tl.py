#!/usr/bin/env python3
"""C1 (2^20 f32, 1-D, Lorenzo + regression in blocks of 128; or 'default': the tuner's Lorenzo-1 / Lorenzo-2 set) decompress steps for a timeline"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field1d
n = 1 << 20
a = field1d(n, np.float32); dev = torch.device("cuda:0"); t = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(n); conf.cmprAlgo = sz3_amd.ALGO_INTERP_LORENZO if len(sys.argv) > 1 and sys.argv[1] == "default" else sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(n, np.float32)
cap = max(dc.payload_bound(n), dc.payload_bound_conf(conf)); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
out = torch.empty_like(t)
for _ in range(4): dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
torch.cuda.synchronize()
print("C1 decompress %s: %.4f ms/call err %.3g" % (sys.argv[1:] or ["composed"], (time.perf_counter() - t0) / 20 * 1e3, float((out - t).abs().max())))
