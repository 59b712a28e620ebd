#!/usr/bin/env python3
"""two contexts in flight on two streams (a producer compressing a series of fields): stage 1 of one call beside stage 2 of the other.
C2 workload; prints ms per call for one context (the bench's step) and for two alternating ones."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = 512; dev = torch.device("cuda:0")
arrs = [torch.from_numpy(field3d((S, S, S), seed=20260928 + k)).to(dev) for k in range(2)]
conf = sz3_amd.Config(S, S, S); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = 1e-3
n = S ** 3
dcs = [sz3_amd.DeviceCompressor(n, np.float32) for _ in range(2)]
cap = dcs[0].payload_bound(n)
pls = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
def one(k, steps):
    st = streams[k].cuda_stream
    for _ in range(steps):
        dcs[k].stage1(conf, arrs[k].data_ptr(), st); dcs[k].stage2(pls[k].data_ptr(), cap, st); dcs[k].finish(st)
def two(steps):
    pending = [False, False]
    for i in range(steps):
        k = i & 1; st = streams[k].cuda_stream
        if pending[k]: dcs[k].finish(st)
        dcs[k].stage1(conf, arrs[k].data_ptr(), st); dcs[k].stage2(pls[k].data_ptr(), cap, st); pending[k] = True
    for k in range(2):
        if pending[k]: dcs[k].finish(streams[k].cuda_stream)
for k in range(2): one(k, 4)
torch.cuda.synchronize(); t0 = time.perf_counter(); one(0, 40); torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 40 * 1e3
two(8); torch.cuda.synchronize(); t0 = time.perf_counter(); two(80); torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 80 * 1e3
print("one context: %.4f ms per call (%.0f GB/s); two contexts in flight: %.4f ms per call (%.0f GB/s)" % (t1, n * 4 / t1 / 1e6, t2, n * 4 / t2 / 1e6))
print("speculation", dcs[0].spec_stats(), dcs[1].spec_stats())
