#!/usr/bin/env python3
"""decompress steps of a C2-like Lorenzo stream for a timeline: tools/r6/dec_lab.py z,y,x [flags]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,512,512").split(","))
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
t = torch.from_numpy(field3d(shape)).to(dev)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = 1e-3
n = t.numel(); dc = sz3_amd.DeviceCompressor(n, np.float32)
cap = dc.payload_bound(n); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
for _ in range(3): size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
out = torch.empty_like(t)
sz3_amd.lib().sz3hip_debug_flags(flags)
for _ in range(4): dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("decompress %s flags %d: %.4f ms/call, err %.3g" % (shape, flags, dt * 1e3, float((out - t).abs().max())))
