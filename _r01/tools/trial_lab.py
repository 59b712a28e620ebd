#!/usr/bin/env python3
"""per-pass timing of one tuner trial workgroup (k_interp_trials, workgroup (block 0, trial 1))"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512")); eb = float(os.environ.get("LAB_EB", "1e-4"))
a = field3d((S, S, S)); dev = torch.device("cuda:0"); d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(S, S, S); conf.absErrorBound = eb
dc = sz3_amd.DeviceCompressor(a.size, np.float32); cap = dc.payload_bound(a.size)
pl = torch.empty(cap, dtype=torch.uint8, device=dev)
for _ in range(2): dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, 0)
torch.cuda.synchronize()
ts = (C.c_uint64 * 72)(); L = sz3_amd.lib(); L.szk_debug_trial_ts(ts)
np_ = ts[71]
print("passes", np_)
for k in range(np_ + 1): print("  %2d %7.2f us" % (k, (ts[k + 1] - ts[k]) / 100.0))
print("total %.1f us" % ((ts[np_ + 1] - ts[0]) / 100.0))
