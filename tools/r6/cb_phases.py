#!/usr/bin/env python3
"""phase timing of the wide code book kernel (wall_clock64 stamps, 100 MHz) at C1 / C3 / a 2-D field: tools/r6/cb_phases.py c1|c3|2d"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d, field1d, field2d
case = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
if case == "c1":
    a = field1d(1 << 20); conf = sz3_amd.Config(a.size); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound = 1e-3
elif case == "2d":
    a = field2d((4096, 4096)); conf = sz3_amd.Config(*a.shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound = 1e-5; conf.regression = 0
else:
    a = field3d((512, 512, 512)); conf = sz3_amd.Config(*a.shape); conf.cmprAlgo = sz3_amd.ALGO_INTERP; conf.absErrorBound = 1e-4
d_in = torch.from_numpy(a).to(dev)
if os.environ.get("SZ3_LAB_FLAGS"): sz3_amd.lib().sz3hip_debug_flags(int(os.environ["SZ3_LAB_FLAGS"]))
dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
for _ in range(3): n = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, 0)
L = sz3_amd.lib(); L.sz3hip_debug_codebook_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 16)(); L.sz3hip_debug_codebook_info(dc._h, out)
ts = [out[4 + i] for i in range(9)]
print("%s: n_symbols %d max_len %d sym_min %d sym_count %d; payload %d" % ((case,) + tuple(out[:4]) + (n,)))
names = ["(entry)", "(entry)", "compact/class", "sort", "merge", "depth+lengths", "scatter", "lens by key", "tail"]
ts12 = [out[4 + i] for i in range(12)]
print("  ts[1]-ts[0] zeroing %.2f us; ts[9]-ts[1] octave counts %.2f us; ts[2]-ts[9] class compaction %.2f us" % ((ts12[1]-ts12[0])/100.0, (ts12[9]-ts12[1])/100.0, (ts12[2]-ts12[9])/100.0))
for i in range(2, 9): print("  ts[%d] %-14s +%7.2f us" % (i, names[i], (ts[i] - ts[i - 1]) / 100.0 if ts[i - 1] else 0.0))
