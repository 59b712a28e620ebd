#!/usr/bin/env python3
"""host API end to end at C2, five calls each way, the library's own timing on stderr (SZ3HIP_TIMING=1); arrays are kept alive so that
no munmap of an earlier call's array falls into a timed call"""
import os, sys, time
os.environ.setdefault("SZ3HIP_TIMING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512")); eb = float(os.environ.get("LAB_EB", "1e-3"))
a = field3d((S, S, S))
conf = sz3_amd.Config(S, S, S); conf.absErrorBound = eb
conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0
keep = []
for it in range(5):
    t0 = time.perf_counter(); blob, ratio = sz3_amd.compress(a, conf); t1 = time.perf_counter()
    keep.append(blob)
    print("iter %d: compress %.1f ms (%.2f GB/s) ratio %.3f" % (it, (t1 - t0) * 1e3, a.nbytes / (t1 - t0) / 1e9, ratio), file=sys.stderr)
for it in range(5):
    t1 = time.perf_counter(); dec, c2 = sz3_amd.decompress(blob, a.dtype, a.shape); t2 = time.perf_counter()
    keep.append(dec)
    print("iter %d: decompress %.1f ms (%.2f GB/s)" % (it, (t2 - t1) * 1e3, a.nbytes / (t2 - t1) / 1e9), file=sys.stderr)
assert abs(dec.astype(np.float64) - a).max() <= eb
