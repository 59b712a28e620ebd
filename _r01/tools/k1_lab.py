#!/usr/bin/env python3
"""Ablation timings of the stage-1 kernel (results are wrong for flags != 0). Usage: python tools/k1_lab.py [flags...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512"))
a = field3d((S, S, S)); dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(S, S, S); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(a.size, np.float32)
stream = torch.cuda.current_stream().cuda_stream
L = sz3_amd.lib(); L.sz3hip_debug_flags.argtypes = [__import__("ctypes").c_int]
flags = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 16, 1 | 2, 1 | 4, 1 | 2 | 4]
for f in flags:
    L.sz3hip_debug_flags(f)
    for _ in range(3): dc.stage1(conf, d_in.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dc.stage1(conf, d_in.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    print("flags %2d: stage1 %.1f us (incl. 2 memsets + hist_reduce)" % (f, 1e3 * e0.elapsed_time(e1) / 10))
L.sz3hip_debug_flags(0)
