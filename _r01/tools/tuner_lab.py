#!/usr/bin/env python3
"""one ALGO_INTERP_LORENZO compress at 512^3 (run under rocprofv3 --kernel-trace to see the tuner's kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512")); eb = float(os.environ.get("LAB_EB", "1e-4"))
a = field3d((S, S, S)); dev = torch.device("cuda:0"); d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(S, S, S); conf.absErrorBound = eb
dc = sz3_amd.DeviceCompressor(a.size, np.float32); cap = dc.payload_bound(a.size)
pl = torch.empty(cap, dtype=torch.uint8, device=dev)
for _ in range(2): n = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, 0)
torch.cuda.synchronize(); print("payload", n, dc.tuner_report())
