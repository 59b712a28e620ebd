#!/usr/bin/env python3
"""timings of the stock-stream paths at 512^3 f32, abs 1e-4 (C3's field): this library's own container (id 17), the
stock container written here (SZ3HIP_TIMING=1 prints the stages), and a stock stream read back"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512"))
a = field3d((S, S, S))
conf = sz3_amd.Config(S, S, S); conf.absErrorBound = 1e-4
for stock in (0, 1):
    sz3_amd.set_stock_format(bool(stock))
    for rep in range(2):
        t0 = time.perf_counter(); blob, ratio = sz3_amd.compress(a, conf); t1 = time.perf_counter()
        dec, c2 = sz3_amd.decompress(blob, np.float32, a.shape); t2 = time.perf_counter()
    err = float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))))
    print("stock_format %d: algo %d ratio %.3f  compress %.1f ms (%.2f GB/s)  decompress %.1f ms (%.2f GB/s)  max err %.3g" % (
        stock, c2.cmprAlgo, ratio, 1e3 * (t1 - t0), a.nbytes / (t1 - t0) / 1e9, 1e3 * (t2 - t1), a.nbytes / (t2 - t1) / 1e9, err), flush=True)
sz3_amd.set_stock_format(False)
