#!/usr/bin/env python3
"""phase timing of the codebook kernel (wall_clock64 stamps, 100 MHz)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "256"))
shape = tuple(int(v) for v in os.environ['LAB_SHAPE'].split(',')) if os.environ.get('LAB_SHAPE') else (S, S, S)
dt = np.float64 if os.environ.get('LAB_DTYPE') == 'f64' else np.float32
a = field3d(shape, dt, sigma=2e-6) if dt == np.float64 else field3d(shape); dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
if os.environ.get('SZ3_LAB_FLAGS'): sz3_amd.lib().sz3hip_debug_flags(int(os.environ['SZ3_LAB_FLAGS']))
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_INTERP if os.environ.get('LAB_ALGO') == 'interp' else sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound = float(os.environ.get("LAB_EB", "1e-3")); conf.regression = 0
dc = sz3_amd.DeviceCompressor(a.size, dt)
cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
for _ in range(3): dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, 0)
L = sz3_amd.lib(); L.sz3hip_debug_codebook_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 16)(); L.sz3hip_debug_codebook_info(dc._h, out)
ts = [out[4 + i] for i in range(9)]
ts[1] = ts[0]  # (no separate sweep phase any more)
print("n_symbols %d max_len %d sym_min %d sym_count %d" % tuple(out[:4]))
names = ["sweep", "compact", "sort", "merge", "depth", "lengths", "scatter", "assign"]
for i, n in enumerate(names): print("  %-8s %7.2f us" % (n, (ts[i + 1] - ts[i]) / 100.0))
print("  total    %7.2f us" % ((ts[8] - ts[0]) / 100.0))
