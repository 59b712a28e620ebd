#!/usr/bin/env python3
"""device compress / decompress times of a few shapes the bench does not cover; run once per build tree: AB_ROOT=<tree> python tools/ab_cases.py"""
import os, sys, time
ROOT = os.environ.get("AB_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d, field4d, field2d
dev = torch.device("cuda:0")
ONLY = os.environ.get("AB_ONLY")
def run(name, a, algo, eb):
    if ONLY and not name.startswith(ONLY): return
    t = torch.from_numpy(a).to(dev)
    conf = sz3_amd.Config(*a.shape); conf.cmprAlgo = algo; conf.absErrorBound = eb
    if algo == sz3_amd.ALGO_LORENZO_REG: conf.regression = 0
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev); out = torch.empty_like(t)
    ts, td = [], []
    for it in range(6):
        torch.cuda.synchronize(); e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e0.record(); n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0); e1.record()
        dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0); e2.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)); td.append(e1.elapsed_time(e2))
    print("%-28s compress %8.3f ms  decompress %8.3f ms  ratio %.3f" % (name, min(ts[2:]), min(td[2:]), a.nbytes / n), flush=True)
LZ, IP, DF = sz3_amd.ALGO_LORENZO_REG, sz3_amd.ALGO_INTERP, sz3_amd.ALGO_INTERP_LORENZO
a4 = field4d((12, 256, 256, 256))
run("4d 12x256^3 lorenzo 1e-3", a4, LZ, 1e-3); run("4d 12x256^3 default 1e-3", a4, DF, 1e-3)
a2 = field2d((8192, 8192)); run("2d 8192^2 lorenzo 1e-3", a2, LZ, 1e-3); run("2d 8192^2 interp 1e-3", a2, IP, 1e-3)
a1 = field3d((512, 512, 512)).reshape(-1); run("1d 2^27 lorenzo 1e-3", a1, LZ, 1e-3)
an = field3d((512, 512, 512)); an.reshape(-1)[::50] = np.nan
run("3d 512^3 2% NaN lorenzo", an, LZ, 1e-3)
a5 = field3d((13, 500, 500)); run("3d 13x500x500 default 1e-3", a5, DF, 1e-3)
a6 = field3d((500, 500, 500)); run("3d 500^3 lorenzo 1e-3", a6, LZ, 1e-3); run("3d 500^3 default 1e-4", a6, DF, 1e-4)
