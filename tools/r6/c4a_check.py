#!/usr/bin/env python3
"""C4a-like slab (f64, Lorenzo + regression): per-call times and the hand-over speculation's outcome"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field_c4a
shape = (64, 512, 512)
a = field_c4a(shape); dev = torch.device("cuda:0")
t = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound = 1e-6
dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
cap = max(dc.payload_bound(a.size), dc.payload_bound_conf(conf)); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = dc.stats()
    print("call %d: %.3f ms  size %d  spec (hits, misses) %s" % (i, dt * 1e3, n, dc.spec_stats() if hasattr(dc, "spec_stats") else "?"))
