#!/usr/bin/env python3
"""C3 (512^3 f32, default algorithm = tuner + interpolation, abs 1e-4) steps under sz3hip_debug_flags: tools/r6/c3_lab.py <flags> [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shape = (512, 512, 512)
dev = torch.device("cuda:0")
ts = [torch.from_numpy(field3d(shape, seed=sd)).to(dev) for sd in (20260928, 7)]
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_INTERP_LORENZO; conf.absErrorBound = 1e-4
n = ts[0].numel()
dc = sz3_amd.DeviceCompressor(n, np.float32)
cap = dc.payload_bound(n); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
sz3_amd.lib().sz3hip_debug_flags(flags)
for i in range(4): size = dc.compress(conf, ts[i & 1].data_ptr(), pl.data_ptr(), cap, 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps): size = dc.compress(conf, ts[i & 1].data_ptr(), pl.data_ptr(), cap, 0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
out = torch.empty_like(ts[0]); dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0); torch.cuda.synchronize()
err = float((out - ts[(steps - 1) & 1]).abs().max())
print("C3 flags %d: %.4f ms/step ratio %.4f err %.3g" % (flags, dt * 1e3, n * 4 / size, err))
