#!/usr/bin/env python3
"""stage times of the plain Lorenzo path at C4's slab (f64 128x1024x1024, abs 1e-6) under debug flags: LAB_FLAGS='0 131072'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
shape = tuple(int(v) for v in os.environ.get("LAB_SHAPE", "128,1024,1024").split(","))
dt = np.float32 if os.environ.get("LAB_DTYPE") == "f32" else np.float64
a = field3d(shape, dt, sigma=2e-6) if dt == np.float64 else field3d(shape, dt)
dev = torch.device("cuda:0")
t = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = float(os.environ.get("LAB_EB", "1e-6"))
for flag in [int(f) for f in os.environ.get("LAB_FLAGS", "0 131072").split()]:
    sz3_amd.lib().sz3hip_debug_flags(flag)
    dc = sz3_amd.DeviceCompressor(a.size, dt)
    cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    dc.set_profiling(True) if hasattr(dc, "set_profiling") else None
    for it in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0); e1.record(); torch.cuda.synchronize()
        st = dc.stats()
        print("flag %d call %d: %.3f ms  size %d  %s" % (flag, it, e0.elapsed_time(e1), n, st))
sz3_amd.lib().sz3hip_debug_flags(0)
