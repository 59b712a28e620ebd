#!/usr/bin/env python3
"""mass of the code-length tail of a payload (how often the decoder's first-level table misses)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd, szh_ref
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512")); eb = float(os.environ.get("LAB_EB", "1e-4"))
a = field3d((S, S, S)); dev = torch.device("cuda:0"); d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(S, S, S); conf.absErrorBound = eb
if os.environ.get("LAB_ALGO") == "lorenzo": conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
dc = sz3_amd.DeviceCompressor(a.size, np.float32); cap = dc.payload_bound(a.size)
pl = torch.empty(cap, dtype=torch.uint8, device=dev)
n = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, 0); torch.cuda.synchronize()
codes = dc.debug_codes(a.size).astype(np.int64)
hist = np.bincount(codes, minlength=65536)
h, _o, sec = szh_ref.parse(pl[:n].cpu().numpy().tobytes())
lens = sec["lens"]
print("payload", n, "sections", list(sec.keys()))
if lens is not None:
    full = np.zeros(65536, dtype=np.int64); lo = h.get("sym_min", 0); full[lo:lo + len(lens)] = lens
    tot = hist.sum()
    for L in (8, 10, 12, 13, 14, 15, 16):
        print("len > %2d: mass %.5f  symbols %d" % (L, hist[full > L].sum() / tot, int(((full > L) & (hist > 0)).sum())))
