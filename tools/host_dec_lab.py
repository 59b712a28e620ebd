#!/usr/bin/env python3
"""host decompress into a FRESH array: time by prefault threads (SZ3HIP_PREFAULT_THREADS, SZ3HIP_NO_PREFAULT)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| nproc", os.cpu_count())
a = field3d((512, 512, 512))
conf = sz3_amd.Config(512, 512, 512); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = 1e-3
blob, _ = sz3_amd.compress(a, conf)
for _ in range(2): sz3_amd.decompress(blob, np.float32, a.shape)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); dec, _ = sz3_amd.decompress(blob, np.float32, a.shape); t = time.perf_counter() - t0; best = min(best, t); del dec
out = np.empty(a.size, np.float32); out[:] = 0
b2 = 1e9
for _ in range(5):
    t0 = time.perf_counter(); sz3_amd.decompress(blob, np.float32, a.shape, out=out); b2 = min(b2, time.perf_counter() - t0)
print("threads %s noprefault %s: fresh %.2f ms = %.1f GB/s; reused %.2f ms = %.1f GB/s" % (os.environ.get("SZ3HIP_PREFAULT_THREADS", "8"), os.environ.get("SZ3HIP_NO_PREFAULT", "0"), 1e3 * best, a.nbytes / best / 1e9, 1e3 * b2, a.nbytes / b2 / 1e9))
